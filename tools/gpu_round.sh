#!/bin/bash
# One gpurun call: parity tests, bench, rocprofv3 stats + PMC passes. Everything lands in gpurun_out/.
# usage: gpurun --timeout 1700 -- 'bash tools/gpu_round.sh [tag]'
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
set +e
echo "== env" | tee $OUT/steps.log
(rocminfo | grep -E "gfx|Compute Unit" | head -4; nproc; free -g | head -2) > $OUT/env.log 2>&1
echo "== smoke" | tee -a $OUT/steps.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/steps.log
echo "== pytest probe" | tee -a $OUT/steps.log
timeout 300 python -m pytest tests/test_gpu_probe.py -q -m gpu --timeout 120 > $OUT/pytest_probe.log 2>&1; echo "probe rc=$?" | tee -a $OUT/steps.log
echo "== pytest hgemm" | tee -a $OUT/steps.log
timeout 900 python -m pytest tests/test_gpu_hgemm.py -q -m gpu --timeout 300 > $OUT/pytest_hgemm.log 2>&1; echo "hgemm rc=$?" | tee -a $OUT/steps.log
echo "== pytest attn" | tee -a $OUT/steps.log
timeout 900 python -m pytest tests/test_gpu_attn.py -q -m gpu --timeout 300 > $OUT/pytest_attn.log 2>&1; echo "attn rc=$?" | tee -a $OUT/steps.log
echo "== bench" | tee -a $OUT/steps.log
timeout 600 python bench.py --steps 30 --warmup 3 --sweep > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/steps.log
echo "== rocprof stats" | tee -a $OUT/steps.log
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o stats -- python tools/prof_kernels.py --iters 5 > $OUT/prof_stats.log 2>&1; echo "stats rc=$?" | tee -a $OUT/steps.log
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout 400 rocprofv3 --pmc $pass -d $OUT/pmc_$name -o pmc -- python tools/prof_kernels.py --iters 2 > $OUT/pmc_$name.log 2>&1; echo "pmc $name rc=$?" | tee -a $OUT/steps.log
done
# keep only compact artefacts (<= 64 MiB comes back)
find $OUT -name "*.db" -size +20M -delete
du -sh $OUT | tee -a $OUT/steps.log
