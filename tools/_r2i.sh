mkdir -p gpurun_out/r2i
tools/power_watch.sh r2i -- bash -c "timeout 80 tools/cpp/mfma_power.bin --seconds 1.5 --modes 0,16,32,48,10,26; timeout 60 tools/cpp/mfma_power.bin --bits --seconds 1.5 --modes 0,16,48" > gpurun_out/r2i/run.log 2>&1
cat gpurun_out/r2i/run.log
