#!/bin/bash
# One-off PMC passes on the GEMM kernels (ours + hipBLASLt's, same inputs): where do the LDS-DMA loads wait?  Texture-address /
# L1 (TA, TCP) and sequencer (SQ) counters, a few per pass (separate rocprofv3 --pmc runs, no tracing domains).
# usage: tools/pmc_vmem.sh <tag>   -> gpurun_out/<tag>/pmc_vmem.txt
TAG=${1:-r3n}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
set +e
pmc() { local name=$1; shift; timeout 300 rocprofv3 --pmc "$@" -d $OUT/pmcv_$name -o pmc -- python tools/prof_kernels.py --what hgemm --iters 2 > $OUT/pmcv_$name.log 2>&1; echo "pmc $name rc=$?" | tee -a $OUT/steps.log; }
pmc ta1 TA_BUFFER_TOTAL_CYCLES TA_BUFFER_COALESCED_READ_CYCLES
pmc ta2 TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES
pmc ta3 TA_BUFFER_READ_LDS_WAVEFRONTS TA_BUFFER_COALESCEABLE_WAVEFRONTS
pmc tcp1 TCP_TOTAL_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES
pmc tcp2 TCP_TCC_READ_REQ_LATENCY TCP_TCP_LATENCY TCP_READ_TAGCONFLICT_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES
pmc sq1 SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS
pmc sq2 SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU
python - "$OUT" > $OUT/pmc_vmem.txt 2>&1 <<'PY'
import sqlite3, sys, glob, collections
out = sys.argv[1]
tab = collections.defaultdict(dict)
for db in sorted(glob.glob(out + "/pmcv_*/**/*.db", recursive=True)):
    cur = sqlite3.connect(db).cursor()
    for name, cname, val, cnt in cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                                              "where kernel_name like '%hgemm_w4y%' or kernel_name like '%hgemm_w4x%' or kernel_name like '%Cijk_%' group by kernel_name, counter_name"):
        tab[cname][name[:60]] = (val, cnt)
kern = sorted({k for c in tab.values() for k in c})
for i, k in enumerate(kern):
    print(f"K{i} = {k}")
print(f"{'counter':40s}" + "".join(f"{'K%d' % i:>16s}" for i in range(len(kern))))
for c in sorted(tab):
    print(f"{c:40s}" + "".join(f"{tab[c].get(k, (float('nan'), 0))[0]:16.4g}" for k in kern))
PY
find $OUT -name "*.db" -delete
cat $OUT/pmc_vmem.txt
