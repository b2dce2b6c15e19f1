mkdir -p gpurun_out/r2l
tools/power_watch.sh r2l -- bash -c "timeout 100 tools/cpp/mfma_power.bin --seconds 1.5 --modes 0,1,16,10,11,26,4,14,15; timeout 60 tools/cpp/mfma_power.bin --zero --seconds 1.5 --modes 0,1,10,14" > gpurun_out/r2l/run.log 2>&1
cat gpurun_out/r2l/run.log
