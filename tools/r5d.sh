#!/bin/bash
# round 5, fourth GPU call: N % 256 == 128 on the merged-phase kernel, the new full-size parity tests, kernel durations of a small split-KV launch
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r5d && export TMPDIR=/tmp
O=gpurun_out/r5d
timeout 1500 python -m pytest tests -m gpu -x -q -k "n_multiple_of_128 or d256_bf16 or mid_size or split_kv or non_finite or workgroup_shapes" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout 600 python tools/attn_rate.py --seconds 0.4 --rounds 3 \
  4,32,4224,128 4,32,4224,128:nw=8 4,32,4224,128:nw=4 4,32,4096,128 1,48,8320,64 1,48,8320,64:nw=4 1,48,8192,64 \
  4,32,1152,128 4,32,1152,128:nw=4 2,16,896,64 2,16,896,64:nw=4 4,32,4224,128:vt 4,32,4224,128:vt:nw=4 > $O/attn_n128.log 2>&1; cat $O/attn_n128.log
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_small -o small -- python $GRAFT_REPO_ROOT/tools/attn_rate.py --seconds 0.05 --rounds 1 1,8,1024,128:split=1 1,8,1024,128 1,8,1024,128:split_fuse=1 1,4,4096,128:split=1 1,4,4096,128 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find /tmp/prof_small -name "*kernel_stats.csv" | head -1); echo "stats file: $f"; python - "$f" > $O/small_split_kernel_stats.log <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("# rocprofv3 --kernel-trace --stats of tools/attn_rate.py 1,8,1024,128 (split off / auto = 4 / fused) and 1,4,4096,128 (off / auto = 4)")
for r in rows:
    if "attn" in r["Name"]:
        print(f'{r["Name"][:110]:110s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:8.2f} us  min {float(r["MinNs"])/1e3:8.2f}  max {float(r["MaxNs"])/1e3:8.2f}')
P
cat $O/small_split_kernel_stats.log
