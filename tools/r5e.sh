#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r5e && export TMPDIR=/tmp
O=gpurun_out/r5e
rocm-smi --showuse 2>/dev/null | head -8
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 600 python -m pytest tests/test_gpu_attn.py -m gpu -x -q -k "workgroup_shapes" > $O/pytest1.log 2>&1; tail -3 $O/pytest1.log
timeout 600 python -m pytest tests/test_gpu_attn.py -m gpu -x -q -k "n_multiple_of_128" > $O/pytest2.log 2>&1; tail -3 $O/pytest2.log
