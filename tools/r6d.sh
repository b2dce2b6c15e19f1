#!/bin/bash
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r6d; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_hgemm.py -q -x > $OUT/pytest_hgemm.log 2>&1; tail -5 $OUT/pytest_hgemm.log
timeout 600 python -m pytest tests/test_gpu_attn.py -q -x -k "config3 or agree_with_each_other" > $OUT/pytest_new.log 2>&1; tail -5 $OUT/pytest_new.log
timeout 900 python tools/hgemm_sizes.py sweep $OUT/sweep_small.json 256 4352 256 0.3 > $OUT/sweep_small.log 2>&1
grep -v amdgpu.ids $OUT/sweep_small.log
