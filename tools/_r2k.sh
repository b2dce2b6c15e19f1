mkdir -p gpurun_out/r2k
rocm-smi --showclocks > gpurun_out/r2k/clocks_idle.txt 2>&1
tools/power_watch.sh r2k -- bash -c "timeout 80 tools/cpp/mfma_power.bin --seconds 1.5 --modes 0,4,14; timeout 60 tools/cpp/mfma_power.bin --zero --seconds 1.5 --modes 0; python tools/sustain.py --seconds 1.5 hgemm:abl=14 hgemm:abl=14:zero hgemm hgemm:zero" > gpurun_out/r2k/run.log 2>&1
cat gpurun_out/r2k/run.log
