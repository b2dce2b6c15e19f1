#!/bin/bash
# round 5, first GPU call: GPU tests, smoke, the new HGEMM shapes and split-KV rates, the driver-style bench line
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r5a && export TMPDIR=/tmp
O=gpurun_out/r5a
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 600 python tools/hgemm_shapes.py --seconds 0.6 8192,8192,8192:auto:vendor 8320,8320,8320:auto:mfma128:vendor 8192,8320,8192:auto:vendor 8320,8192,8192:auto:vendor 8192,8192,8224:auto:mfma128:vendor 4224,4224,4128:auto:mfma128:vendor > $O/hgemm_shapes.log 2>&1; cat $O/hgemm_shapes.log
timeout 600 python tools/attn_rate.py --seconds 0.3 --rounds 3 \
  1,8,1024,128:split=1 1,8,1024,128:split=2 1,8,1024,128:split=4 1,8,1024,128:split=8 1,8,1024,128 \
  1,8,2048,64:split=1 1,8,2048,64:split=2 1,8,2048,64:split=4 1,8,2048,64:split=8 1,8,2048,64 \
  1,16,2048,128:split=1 1,16,2048,128:split=2 1,16,2048,128:split=4 1,16,2048,128 1,16,2048,128:split=1:nw=4 \
  1,32,1024,128:split=1 1,32,1024,128:split=2 1,32,1024,128 1,32,1024,64:split=1 1,32,1024,64:split=2 \
  1,4,4096,128:split=1 1,4,4096,128:split=4 1,4,4096,128:split=8 1,4,4096,128:split=16 1,4,4096,128 \
  1,48,2048,64:split=1 1,48,2048,64:split=2 > $O/attn_split.log 2>&1; cat $O/attn_split.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; python - <<'P'
import json
d=json.load(open('gpurun_out/r5a/bench_default.json'))
print({k:d[k] for k in ('value','ms_per_step','scaling','n_gpus')}); print(d['headline']); print({k:v for k,v in d['roofline'].items() if k.startswith('attn_cfg3')})
P
