#!/bin/bash
# round 5, eighth GPU call: D = 1024 KV-walk stagger A/B
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r5h && export TMPDIR=/tmp
O=gpurun_out/r5h
timeout 900 python -m pytest tests/test_gpu_attn.py tests/test_gpu_hgemm.py tests/test_gpu_fullsize.py -m gpu -x -q -k "bigd_block_map or mfma128 or d1024 or large_head" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
timeout 600 python tools/attn_rate.py --seconds 0.5 --rounds 3 1,48,8192,1024 1,48,8192,1024:bigd_stagger=1 1,48,8192,1024:bigd_map=1 1,48,8192,1024:bigd_map=1:bigd_stagger=1 1,48,4096,1024 1,48,4096,1024:bigd_stagger=1 > $O/bigd_stagger.log 2>&1; cat $O/bigd_stagger.log
