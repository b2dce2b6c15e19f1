mkdir -p gpurun_out/r2n
timeout 300 python -m pytest tests/test_gpu_hgemm.py -m gpu -x -q > gpurun_out/r2n/pytest.log 2>&1; tail -5 gpurun_out/r2n/pytest.log
tools/power_watch.sh r2n -- bash -c "python tools/sustain.py --seconds 2.5 hgemm:var=w4c hgemm:var=w4x vendor hgemm:var=w4x:zero hgemm:var=w4x:uniform hgemm:var=w4c" > gpurun_out/r2n/run.log 2>&1
cat gpurun_out/r2n/run.log
