#!/bin/bash
# One gpurun call. usage: gpurun --timeout 1700 -- 'bash tools/gpu_round.sh <tag> "<steps>"'
# steps: smoke probe hgemm attn bench stats pmc_fetch pmc_write pmc_mfma pmc_lds   (default: all)
TAG=${1:-r01}
STEPS=${2:-"smoke probe hgemm attn bench stats pmc_fetch pmc_write pmc_mfma pmc_lds"}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
set +e
has() { [[ " $STEPS " == *" $1 "* ]]; }
log() { echo "$@" | tee -a $OUT/steps.log; }
(rocminfo | grep -E "gfx|Compute Unit" | head -4; nproc; free -g | head -2) > $OUT/env.log 2>&1
if has smoke; then timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; log "smoke rc=$?"; fi
if has probe; then timeout 300 python -m pytest tests/test_gpu_probe.py -q -m gpu --timeout 120 > $OUT/pytest_probe.log 2>&1; log "probe rc=$?"; fi
if has hgemm; then timeout 900 python -m pytest tests/test_gpu_hgemm.py -q -m gpu --timeout 300 > $OUT/pytest_hgemm.log 2>&1; log "hgemm rc=$?"; fi
if has attn; then timeout 900 python -m pytest tests/test_gpu_attn.py -q -m gpu --timeout 300 > $OUT/pytest_attn.log 2>&1; log "attn rc=$?"; fi
if has bench; then timeout 600 python bench.py --steps 30 --warmup 3 --sweep > $OUT/bench.json 2> $OUT/bench.err; log "bench rc=$?"; fi
if has stats; then timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o stats -- python tools/prof_kernels.py --iters 5 > $OUT/prof_stats.log 2>&1; log "stats rc=$?"; fi
pmc() { # name, counters...
  local name=$1; shift
  timeout 400 rocprofv3 --pmc "$@" -d $OUT/pmc_$name -o pmc -- python tools/prof_kernels.py --iters 2 > $OUT/pmc_$name.log 2>&1; log "pmc $name rc=$?"
}
if has pmc_fetch; then pmc fetch FETCH_SIZE; fi
if has pmc_write; then pmc write WRITE_SIZE; fi
if has pmc_mfma; then pmc mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE; fi
if has pmc_lds; then pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY; fi
find $OUT -name "*.db" -size +20M -delete
du -sh $OUT | tee -a $OUT/steps.log
