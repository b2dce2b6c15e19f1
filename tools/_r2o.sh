mkdir -p gpurun_out/r2o
timeout 300 python -m pytest tests/test_gpu_hgemm.py -m gpu -x -q > gpurun_out/r2o/pytest.log 2>&1; tail -5 gpurun_out/r2o/pytest.log
tools/power_watch.sh r2o -- bash -c "python tools/sustain.py --seconds 2.5 hgemm:var=w4c hgemm:var=w4x hgemm:var=w4y vendor hgemm:var=w4y:zero vendor:zero hgemm:var=w4y" > gpurun_out/r2o/run.log 2>&1
cat gpurun_out/r2o/run.log
