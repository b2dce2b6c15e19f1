#!/bin/bash
# TA / TCP / SQ counters of the attention kernels (separate rocprofv3 --pmc passes of tools/prof_kernels.py --what attn).
# usage: tools/pmc_attn.sh <tag>   -> gpurun_out/<tag>/pmc_attn.txt
TAG=${1:-r3u}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
set +e
pmc() { local name=$1; shift; timeout 400 rocprofv3 --pmc "$@" -d $OUT/pmca_$name -o pmc -- python tools/prof_kernels.py --what attn --iters 2 > $OUT/pmca_$name.log 2>&1; echo "pmc $name rc=$?" | tee -a $OUT/steps.log; }
pmc lat TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_TCP_LATENCY TCP_PENDING_STALL_CYCLES
pmc ta TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES
pmc sq SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
pmc sq2 SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY
pmc sq3 SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_BUSY_CYCLES
pmc mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES
python - "$OUT" > $OUT/pmc_attn.txt 2>&1 <<'PY'
import sqlite3, sys, glob, collections, re
out = sys.argv[1]
tab = collections.defaultdict(dict)
for db in sorted(glob.glob(out + "/pmca_*/**/*.db", recursive=True)):
    cur = sqlite3.connect(db).cursor()
    for name, cname, val, cnt in cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                                              "where kernel_name like '%attn_fwd%' or kernel_name like '%gemm_fp8%' group by kernel_name, counter_name"):
        m = re.search(r"\d+(attn_fwd_\w+?_kernel|attn_fwd_kernel|gemm_fp8_\w+?_kernel)I?([^E]*)E", name)
        k = (m.group(1) + "<" + re.sub(r"L[ib]", "", m.group(2)) + ">") if m else name[:40]
        tab[cname][k] = val
kern = sorted({k for c in tab.values() for k in c})
for i, k in enumerate(kern):
    print(f"K{i} = {k}")
print(f"{'counter':34s}" + "".join(f"{'K%d' % i:>13s}" for i in range(len(kern))))
for c in sorted(tab):
    print(f"{c:34s}" + "".join(f"{tab[c].get(k, float('nan')):13.4g}" for k in kern))
PY
find $OUT -name "*.db" -delete
cat $OUT/pmc_attn.txt
