#!/bin/bash
TAG=r3ae; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cat > /tmp/w8g_run.py <<'PY'
import sys
sys.path.insert(0, ".")
import torch
from leetcuda_amd import capi, host
capi.load()
q, k, v, o, tv = host.get_qkvo(1, 48, 8192, 64, seed=0)
for nw in (513, 516):
    capi.tune("attn_nw", nw)
    for _ in range(3):
        capi.attn_fwd(q, k, v, o)
torch.cuda.synchronize()
PY
pmc() { local name=$1; shift; timeout 200 rocprofv3 --pmc "$@" -d $OUT/pmc8_$name -o pmc -- python /tmp/w8g_run.py > $OUT/pmc8_$name.log 2>&1; echo "pmc $name rc=$?"; }
pmc mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES
pmc sq SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_ANY
pmc sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LEVEL_WAVES
python - "$OUT" <<'PY'
import sqlite3, sys, glob, collections
out = sys.argv[1]
tab = collections.defaultdict(dict)
for db in sorted(glob.glob(out + "/pmc8_*/**/*.db", recursive=True)):
    cur = sqlite3.connect(db).cursor()
    for name, cname, val, cnt in cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%attn_fwd%' group by kernel_name, counter_name"):
        tab[cname]["w8g" if "w8g" in name else "w4g"] = val
print(f"{'counter':32s}{'w4g':>14s}{'w8g':>14s}")
for c in sorted(tab):
    print(f"{c:32s}{tab[c].get('w4g', float('nan')):14.4g}{tab[c].get('w8g', float('nan')):14.4g}")
PY
find $OUT -name "*.db" -delete
