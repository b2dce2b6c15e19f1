#!/bin/bash
# round 5, seventh GPU call: the eight-wave 128-tile kernel (intra-workgroup split-K) — parity + rates at the sizes LC_HGEMM_AUTO sends to it
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r5g && export TMPDIR=/tmp
O=gpurun_out/r5g
timeout 900 python -m pytest tests/test_gpu_hgemm.py tests/test_gpu_fullsize.py -m gpu -x -q -k "mfma128 or flagship or border or legal or mid_size or golden or reference_entry" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
for w in 1 2; do echo "== hgemm_128w = $w"; LC_W=$w timeout 300 python - <<'P' 2>&1 | grep "^n="
import os, sys
sys.path.insert(0, '.')
from leetcuda_amd import capi
capi.load(); capi.tune("hgemm_128w", int(os.environ["LC_W"]))
sys.argv = ["x", "1024,1536,2048,2560,3072,3584", "mfma128", "0.3"]
exec(open("tools/hgemm_sizes.py").read())
P
done > $O/hgemm_128w.log 2>&1; cat $O/hgemm_128w.log
timeout 300 python tools/hgemm_sizes.py 1024,2048,2560,3072,4096 auto 0.4 > $O/hgemm_sizes_auto.log 2>&1; cat $O/hgemm_sizes_auto.log
for ks in 1 0; do echo "== hgemm_splitk = $ks (1 = off: the border blocks run the eight-wave form)"; LC_KS=$ks timeout 200 python - <<'P' 2>&1 | grep SHAPE
import os, sys
sys.path.insert(0, '.')
from leetcuda_amd import capi
capi.load(); capi.tune("hgemm_splitk", int(os.environ["LC_KS"]))
sys.argv = ["x", "--seconds", "0.4", "8192,8320,8192:auto", "8320,8320,8320:auto"]
exec(open("tools/hgemm_shapes.py").read())
P
done > $O/hgemm_border_w8.log 2>&1; cat $O/hgemm_border_w8.log
