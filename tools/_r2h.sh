mkdir -p gpurun_out/r2h
tools/power_watch.sh r2h -- bash -c "python tools/sustain.py --seconds 2 hgemm hgemm:zero hgemm:abl=2 hgemm:abl=2:zero hgemm:abl=8 hgemm:abl=8:zero hgemm:abl=14 hgemm:abl=14:zero hgemm:abl=4 vendor vendor:zero hgemm:uniform; timeout 60 tools/cpp/mfma_power.bin --seconds 1.5 --modes 4,14 --streams 256; timeout 60 tools/cpp/mfma_power.bin --seconds 1.5 --modes 4,14 --streams 32 --region-mib 64" > gpurun_out/r2h/run.log 2>&1
cat gpurun_out/r2h/run.log
