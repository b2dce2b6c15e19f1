mkdir -p gpurun_out/r2m
tools/power_watch.sh r2m -- bash -c "timeout 60 tools/cpp/mfma_power.bin --seconds 1.5 --modes 4,5 --streams 8 --region-kib 512; timeout 60 tools/cpp/mfma_power.bin --seconds 1.5 --modes 4,5; timeout 60 tools/cpp/mfma_power.bin --seconds 1.5 --modes 4,5 --streams 32 --region-mib 64" > gpurun_out/r2m/run.log 2>&1
cat gpurun_out/r2m/run.log
