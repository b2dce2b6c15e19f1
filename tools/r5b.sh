#!/bin/bash
# round 5, second GPU call: split-K border strips, cost-model split-KV rule + cached workspace
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r5b && export TMPDIR=/tmp
O=gpurun_out/r5b
timeout 1500 python -m pytest tests/test_gpu_hgemm.py tests/test_gpu_attn.py tests/test_gpu_fullsize.py -m gpu -x -q -k "flagship or split or border or legal or mfma128 or auto_routes or capture" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python tools/hgemm_shapes.py --seconds 0.6 8192,8192,8192:auto:vendor 8320,8320,8320:auto:vendor 8192,8320,8192:auto:vendor 8320,8192,8192:auto:vendor 8192,8192,8224:auto:vendor 6144,6144,6144:auto:vendor 4224,4224,4128:auto:vendor > $O/hgemm_shapes.log 2>&1; cat $O/hgemm_shapes.log
for ks in 1 2 4 8; do LC_KS=$ks timeout 200 python - <<'P' 2>&1 | grep SHAPE
import os, sys, subprocess
sys.path.insert(0, '.')
from leetcuda_amd import capi
capi.load(); capi.tune("hgemm_splitk", int(os.environ["LC_KS"]))
sys.argv = ["x", "--seconds", "0.4", "8192,8320,8192:auto", "8320,8320,8320:auto"]
print("hgemm_splitk =", os.environ["LC_KS"], flush=True)
exec(open("tools/hgemm_shapes.py").read())
P
done > $O/hgemm_splitk_sweep.log 2>&1; cat $O/hgemm_splitk_sweep.log
timeout 600 python tools/attn_rate.py --seconds 0.3 --rounds 3 \
  1,8,1024,128:split=1 1,8,1024,128:split=2 1,8,1024,128:split=4 1,8,1024,128 \
  1,8,2048,64:split=1 1,8,2048,64:split=4 1,8,2048,64:split=8 1,8,2048,64 \
  1,16,2048,128:split=1 1,16,2048,128 1,32,1024,128 1,32,1024,128:split=2 \
  1,4,4096,128:split=1 1,4,4096,128:split=4 1,4,4096,128:split=8 1,4,4096,128 \
  1,2,8192,128:split=1 1,2,8192,128 1,16,1024,64:split=1 1,16,1024,64 1,24,2048,128:split=1 1,24,2048,128 4,8,512,128:split=1 4,8,512,128 > $O/attn_split.log 2>&1; cat $O/attn_split.log
