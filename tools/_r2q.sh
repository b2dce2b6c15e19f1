mkdir -p gpurun_out/r2q
timeout 300 python -m pytest tests/test_gpu_hgemm.py -m gpu -x -q -k "full_size or w4y" > gpurun_out/r2q/pytest.log 2>&1; tail -3 gpurun_out/r2q/pytest.log
tools/power_watch.sh r2q -- bash -c "python tools/sustain.py --seconds 2.0 hgemm:var=w4y:sched=0 hgemm:var=w4y:sched=1 hgemm:var=w4y:sched=2 vendor hgemm:var=w4y:sched=0:zero hgemm:var=w4y:sched=1:zero hgemm:var=w4y:sched=2:zero hgemm:var=w4y:sched=1 hgemm:var=w4y:sched=2" > gpurun_out/r2q/run.log 2>&1
cat gpurun_out/r2q/run.log
