mkdir -p gpurun_out/r2r
timeout 400 python -m pytest tests/test_gpu_hgemm.py -m gpu -x -q > gpurun_out/r2r/pytest.log 2>&1; tail -5 gpurun_out/r2r/pytest.log
tools/power_watch.sh r2r -- bash -c "python tools/sustain.py --seconds 2.0 hgemm:var=w4c:nn hgemm:var=w4y:nn vendor:nn hgemm:var=w4y:nn:zero hgemm:var=w4y vendor" > gpurun_out/r2r/run.log 2>&1
cat gpurun_out/r2r/run.log
