mkdir -p gpurun_out/r2j
tools/power_watch.sh r2j -- bash -c "timeout 80 tools/cpp/mfma_power.bin --seconds 1.5 --iters 256 --grid-mult 4 --modes 0,32,64,96; timeout 60 tools/cpp/mfma_power.bin --seconds 1.5 --iters 256 --grid-mult 1 --modes 0,96; python tools/sustain.py --seconds 1.5 hgemm:abl=14 hgemm:abl=14:zero" > gpurun_out/r2j/run.log 2>&1
cat gpurun_out/r2j/run.log
