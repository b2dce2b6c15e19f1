#!/bin/bash
# round 5, sixth GPU call: big-head-dim block map A/B (rates + FETCH_SIZE), kernel durations of small split-KV launches, split-KV against wave quantisation
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r5f && export TMPDIR=/tmp
O=$PWD/gpurun_out/r5f; R=$PWD
timeout 900 python -m pytest tests/test_gpu_attn.py -m gpu -x -q -k "bigd_block_map or n_multiple_of_128 or large_head or full_width" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
timeout 600 python tools/attn_rate.py --seconds 0.5 --rounds 3 1,48,8192,1024 1,48,8192,1024:bigd_map=1 1,48,8192,512 1,48,8192,512:bigd_map=1 > $O/bigd_map.log 2>&1; cat $O/bigd_map.log
cat > /tmp/bigd_run.py <<'PY'
import sys; sys.path.insert(0, ".")
import torch
from leetcuda_amd import capi, host
capi.load()
for D in (1024, 512):
    q, k, v, o, _ = host.get_qkvo(1, 48, 8192, D, seed=0)
    for m in (0, 1):
        capi.tune("attn_bigd_map", m)
        for _ in range(2):
            capi.attn_call("flash_attn_mma_stages_split_q_tiling_qkv", q, k, v, o, 2)
        torch.cuda.synchronize()
    capi.tune("attn_bigd_map", 0)
    del q, k, v, o
PY
for c in FETCH_SIZE WRITE_SIZE; do timeout 300 rocprofv3 --pmc $c -d /tmp/pmc_$c -o pmc -- python /tmp/bigd_run.py > $O/pmc_$c.log 2>&1; echo "pmc $c rc=$?"; done
python - > $O/bigd_map_pmc.log 2>&1 <<'PY'
import sqlite3, glob, collections
print("# FETCH_SIZE / WRITE_SIZE (KiB; gfx950: a 128-B read counts 64 B -> fetch x 2) per dispatch, in launch order: map 0, 0, 1, 1 for D = 1024 then D = 512")
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for db in glob.glob(f"/tmp/pmc_{c}/**/*.db", recursive=True):
        cur = sqlite3.connect(db).cursor()
        try:
            rows = cur.execute("select kernel_name, dispatch_id, sum(value) from counters_collection where kernel_name like '%bigd%' group by dispatch_id order by dispatch_id").fetchall()
        except Exception as e:
            print(c, "query failed:", e, [r[0] for r in cur.execute("select name from sqlite_master").fetchall()][:20]); continue
        for name, did, val in rows:
            gb = val * 1024 * (2 if c == "FETCH_SIZE" else 1) / 1e9
            print(f"{c:11s} dispatch {did:4d} {name[:60]:60s} {gb:8.3f} GB")
PY
cat $O/bigd_map_pmc.log
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_small -o small -- python $R/tools/attn_rate.py --seconds 0.05 --rounds 1 1,8,1024,128:split=1 1,8,1024,128 1,8,1024,128:split_fuse=1 1,4,4096,128:split=1 1,4,4096,128 1,4,4096,128:split_fuse=1 > $O/prof_small.log 2>&1; echo "rocprof rc $?"
cd $R; python tools/kernel_durations.py /tmp/prof_small attn > $O/small_split_kernel_durations.log 2>&1; cat $O/small_split_kernel_durations.log
timeout 600 python tools/attn_rate.py --seconds 0.4 --rounds 3 1,24,4096,128:split=1 1,24,4096,128:split=2 1,24,4096,128:split=4 1,40,2048,128:split=1 1,40,2048,128:split=2 \
  1,12,8192,64:split=1 1,12,8192,64:split=2 1,12,8192,64:split=4 1,10,8192,128:split=1 1,10,8192,128:split=2 1,10,8192,128:split=4 1,6,8192,128:split=1 1,6,8192,128 1,6,8192,128:split=4 > $O/attn_split_quant.log 2>&1; cat $O/attn_split_quant.log
