#!/bin/bash
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r6b; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_hgemm.py -q -x -k "one_round" > $OUT/pytest_mid.log 2>&1; tail -5 $OUT/pytest_mid.log
timeout 600 python tools/hgemm_mid_ab.py 1024,1280,1536,1792,2048,2304,2560,2816,3072,3328,3584 0.3 > $OUT/mid_ab.log 2>&1
cat $OUT/mid_ab.log | grep -v amdgpu.ids
timeout 600 python -m pytest tests/test_gpu_attn.py -q -x -k "config3 or agree_with_each_other" > $OUT/pytest_new.log 2>&1; tail -5 $OUT/pytest_new.log
