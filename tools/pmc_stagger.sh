#!/bin/bash
# L2 read latency / TA stalls / fabric bytes of hgemm_w4y under K-loop stagger settings (separate rocprofv3 --pmc passes).
# usage: tools/pmc_stagger.sh <tag> <stagger>...   -> gpurun_out/<tag>/pmc_stagger.txt
TAG=$1; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
set +e
for S in "$@"; do
  pmc() { local name=$1; shift; timeout 300 rocprofv3 --pmc "$@" -d $OUT/pmcs_${S}_$name -o pmc -- python tools/prof_kernels.py --what hgemm --iters 3 --only-auto --stagger $S > $OUT/pmcs_${S}_$name.log 2>&1; echo "pmc $S $name rc=$?" | tee -a $OUT/steps.log; }
  pmc lat TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_TCP_LATENCY TCP_PENDING_STALL_CYCLES
  pmc ta TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES
  pmc fetch FETCH_SIZE
  pmc sq SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
done
python - "$OUT" "$@" > $OUT/pmc_stagger.txt 2>&1 <<'PY'
import sqlite3, sys, glob, collections
out, stags = sys.argv[1], sys.argv[2:]
tab = collections.defaultdict(dict)
for S in stags:
    for db in sorted(glob.glob(f"{out}/pmcs_{S}_*/**/*.db", recursive=True)):
        cur = sqlite3.connect(db).cursor()
        for name, cname, val, cnt in cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                                                  "where kernel_name like '%hgemm_w4y%' or kernel_name like '%Cijk_%' group by kernel_name, counter_name"):
            k = ("ours_nn" if "ILb1E" in name else "ours_tn") if "hgemm_w4y" in name else ("vend_tn" if "Custom" in name else "vend_nn")
            tab[cname][(S, k)] = val
cols = [(S, k) for S in stags for k in ("ours_tn", "ours_nn", "vend_tn")]
print(f"{'counter':34s}" + "".join(f"{(S + ':' + k)[-17:]:>18s}" for S, k in cols))
for c in sorted(tab):
    print(f"{c:34s}" + "".join(f"{tab[c].get(col, float('nan')):18.4g}" for col in cols))
PY
find $OUT -name "*.db" -delete
cat $OUT/pmc_stagger.txt
