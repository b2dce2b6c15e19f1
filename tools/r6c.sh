#!/bin/bash
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r6c; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_hgemm.py -q -x -k "mid_kernel" > $OUT/pytest_mid.log 2>&1; tail -5 $OUT/pytest_mid.log
timeout 900 python tools/hgemm_mid_ab.py 768,1024,1280,1536,1792,2048,2304,2560,2816,3072 0.3 > $OUT/mid_ab.log 2>&1
cat $OUT/mid_ab.log | grep -v amdgpu.ids
timeout 600 python -m pytest tests/test_gpu_attn.py -q -x -k "config3 or agree_with_each_other" > $OUT/pytest_new.log 2>&1; tail -5 $OUT/pytest_new.log
