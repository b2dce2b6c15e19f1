#!/bin/bash
# PMC comparison of the attention schedule variants (clock / MFMA-busy): one --pmc pass + one kernel-trace pass
TAG=${1:-r03k}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o stats -- python tools/attn_variants.py 4 32 4096 128 8,16,32,64,128 > $OUT/stats.log 2>&1; echo "stats rc=$?"
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES -d $OUT/pmc_mfma -o pmc -- python tools/attn_variants.py 4 32 4096 128 8,16,32,64,128 > $OUT/pmc_mfma.log 2>&1; echo "pmc rc=$?"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc_lds -o pmc -- python tools/attn_variants.py 4 32 4096 128 8,16,32,64,128 > $OUT/pmc_lds.log 2>&1; echo "pmc2 rc=$?"
find $OUT -name "*.db" -size +20M -delete
