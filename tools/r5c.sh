#!/bin/bash
# round 5, third GPU call: full GPU suite with the fused split-KV combine, rates, bench line at N = 1 and 2 ranks (gloo, one GPU)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r5c && export TMPDIR=/tmp
O=gpurun_out/r5c
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python tools/attn_rate.py --seconds 0.3 --rounds 3 \
  1,8,1024,128:split=1 1,8,1024,128:split=4:split_fuse=0 1,8,1024,128 1,8,1024,128:split=8 \
  1,8,2048,64:split=1 1,8,2048,64:split_fuse=0 1,8,2048,64 1,8,2048,64:split=8 \
  1,16,2048,128:split=1 1,16,2048,128:split_fuse=0 1,16,2048,128 \
  1,32,1024,128:split=1 1,32,1024,128:split=2 1,32,1024,64:split=1 1,32,1024,64:split=2 \
  1,4,4096,128:split=1 1,4,4096,128 1,4,4096,128:split=8 1,2,8192,128:split=1 1,2,8192,128 1,2,8192,128:split=16 \
  1,8,1024,128:split=1:vt 1,8,1024,128:vt 2,4,512,128:split=1 2,4,512,128:split=2 1,4,1024,64:split=1 1,4,1024,64 > $O/attn_split.log 2>&1; cat $O/attn_split.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
python - <<'P'
import json
d=json.load(open('gpurun_out/r5c/bench_default.json'))
print(d['headline']); print(d['attention_d1024']['roofline'].get('traffic_model'), d['attention_d1024']['roofline'].get('traffic'))
P
