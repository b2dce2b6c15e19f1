#!/bin/bash
# rocprofv3 evidence for profiles/: kernel-trace stats of bench.py ITSELF + separate PMC passes of the hot kernels
TAG=${1:-r02p}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
set +e
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o stats -- python bench.py --steps 20 --warmup 2 --no-cpu-baseline > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err; echo "stats rc=$?" | tee -a $OUT/steps.log
timeout 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline > $OUT/bench_unprofiled.json 2> $OUT/bench_unprofiled.err
pmc() { local name=$1; shift; timeout 400 rocprofv3 --pmc "$@" -d $OUT/pmc_$name -o pmc -- python tools/prof_kernels.py --iters 2 > $OUT/pmc_$name.log 2>&1; echo "pmc $name rc=$?" | tee -a $OUT/steps.log; }
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES
# (round 6) requests the L2 sends to LOCAL MEMORY: the closest thing to a DRAM counter rocprofv3 exposes on gfx950 — it cannot separate Infinity-Cache hits
pmc dram TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum
pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
# summarise ON the box (the databases can exceed what gpurun merges back) and ship the summaries inside gpurun_out/
python tools/summarize_prof.py $TAG > $OUT/summary.log 2>&1; echo "summary rc=$?" | tee -a $OUT/steps.log
cp profiles/${TAG}_* $OUT/ 2>/dev/null
find $OUT -name "*.db" -delete   # (the summaries above are what is kept: gpurun merges back at most 64 MiB)
du -sh $OUT | tee -a $OUT/steps.log
