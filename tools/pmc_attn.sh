#!/bin/bash
# SQ / TA counters of an attention kernel at (1,48,8192,D) (separate rocprofv3 --pmc passes).  usage: tools/pmc_attn.sh <tag> <D> <kernel name pattern> [bf16]
TAG=${1:-r4n}; D=${2:-256}; PAT=${3:-bigd7}; BF=${4:-}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cat > /tmp/bigd4_run.py <<PY
import sys; sys.path.insert(0, ".")
import torch
from leetcuda_amd import capi, host
capi.load()
q, k, v, o, _ = host.get_qkvo(1, 48, 8192, $D, seed=0)
for _ in range(3):
    capi.attn_fwd(q, k, v, o)
torch.cuda.synchronize()
PY
set +e
pmc() { local name=$1; shift; timeout 300 rocprofv3 --pmc "$@" -d $OUT/pmca_$name -o pmc -- python /tmp/bigd4_run.py > $OUT/pmca_$name.log 2>&1; echo "pmc $name rc=$?" | tee -a $OUT/steps.log; }
pmc mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
pmc sq SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
pmc lat TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TA_ADDR_STALLED_BY_TC_CYCLES
pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY
python - "$OUT" "$PAT" > $OUT/pmc_attn_$PAT.txt 2>&1 <<'PY'
import sqlite3, sys, glob, collections
out = sys.argv[1]
tab = collections.defaultdict(float)
for db in sorted(glob.glob(out + "/pmca_*/**/*.db", recursive=True)):
    cur = sqlite3.connect(db).cursor()
    for cname, val in cur.execute("select counter_name, avg(value) from counters_collection where kernel_name like '%" + sys.argv[2] + "%' group by counter_name"):
        tab[cname] = val
for c in sorted(tab):
    print(f"{c:34s}{tab[c]:16.5g}")
w = tab.get("SQ_WAVE_CYCLES", 0)
if w:
    for c in ("SQ_WAIT_INST_ANY", "SQ_VMEM_TA_CMD_FIFO_FULL", "SQ_VMEM_TA_ADDR_FIFO_FULL", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU"):
        if c in tab:
            print(f"{c} / SQ_WAVE_CYCLES = {tab[c] / w:.3f}")
if tab.get("SQ_BUSY_CU_CYCLES"):
    print(f"MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES = {tab['SQ_VALU_MFMA_BUSY_CYCLES'] / tab['SQ_BUSY_CU_CYCLES']:.3f}")
if tab.get("TCP_TCC_READ_REQ"):
    print(f"L2 read latency per request = {tab['TCP_TCC_READ_REQ_LATENCY'] / tab['TCP_TCC_READ_REQ']:.0f} cycles")
PY
find $OUT -name "*.db" -delete
cat $OUT/pmc_attn_$PAT.txt
