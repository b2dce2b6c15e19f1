#!/bin/bash
# f1 on hardware: the reference's UNMODIFIED bench scripts (staged by tools/stage_reference.sh) against this library.
# usage (on the GPU box, from the repo root): tools/run_f1.sh <outdir>
OUT=${1:-gpurun_out/f1}; mkdir -p $OUT; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MPLBACKEND=Agg
S=_refstage/kernels
[ -f $S/hgemm/hgemm.py ] || { echo "no _refstage (run tools/stage_reference.sh first)"; exit 1; }
( cd _refstage && sha256sum -c SHA256SUMS ) > $OUT/f1_sha256_before.txt 2>&1
R="python tools/run_reference_bench.py"
set +e
# 1. every kernel family of hgemm.py at the headline size (hgemm.py:1088-1110 sweeps; --MNK 8192)
timeout 600 $R $S/hgemm/hgemm.py --mma-all --wmma-all --cuda-all --mma-tn --cute-tn --torch --MNK 8192 --sleep 0.02 > $OUT/f1_hgemm_8192_all.log 2>&1; echo "hgemm all rc=$?" | tee -a $OUT/f1_steps.log
# 2. the --plot path (hgemm.py:362-416): MMA families + the two vendor lines over 1024..8192, PNG saved
timeout 600 $R $S/hgemm/hgemm.py --mma-all --mma-tn --cute-tn --MMNK 8192 --SEP 1024 --plot --topk 8 --sleep 0.02 --dir $PWD/$OUT --tag f1_mma > $OUT/f1_hgemm_plot.log 2>&1; echo "hgemm plot rc=$?" | tee -a $OUT/f1_steps.log
# 2b. (round 6) the script's DEFAULT sweep — every M = N = K multiple of 256 up to 12800 (hgemm.py:28-32,419-421), the sweep its README's
#     "98 - 100 % of cuBLAS" is made over — TN families + the cuBLAS (= hipBLASLt) TN line, PNG saved
timeout 1500 $R $S/hgemm/hgemm.py --mma-tn --cute-tn --plot --topk 8 --sleep 0.02 --dir $PWD/$OUT --tag f1_sweep > $OUT/f1_hgemm_default_sweep.log 2>&1; echo "hgemm default sweep rc=$?" | tee -a $OUT/f1_steps.log
# 3. flash_attn_mma.py --check (allclose atol 1e-2 against the flash-attn / SDPA comparators, flash_attn_mma.py:465-494) at
#    config 3 and at the reference's own published shapes
timeout 600 $R $S/flash-attn/flash_attn_mma.py --B 4 --H 32 --N 4096 --D 128 --check --show-all --others --seed 1 > $OUT/f1_fa_cfg3_check.log 2>&1; echo "fa cfg3 rc=$?" | tee -a $OUT/f1_steps.log
timeout 600 $R $S/flash-attn/flash_attn_mma.py --B 1 --H 48 --N 8192 --D 64 --check --show-all --others --seed 1 > $OUT/f1_fa_1x48x8192x64_check.log 2>&1; echo "fa d64 rc=$?" | tee -a $OUT/f1_steps.log
timeout 600 $R $S/flash-attn/flash_attn_mma.py --B 1 --H 8 --N 8192 --D 64 --check --show-all --sdpa --seed 1 > $OUT/f1_fa_1x8x8192x64_check.log 2>&1; echo "fa d64 h8 rc=$?" | tee -a $OUT/f1_steps.log
timeout 600 $R $S/flash-attn/flash_attn_mma.py --B 1 --H 48 --N 8192 --D 512 --check --show-all --sdpa --seed 1 > $OUT/f1_fa_1x48x8192x512_check.log 2>&1; echo "fa d512 rc=$?" | tee -a $OUT/f1_steps.log
# D = 256: the head-dim limit of the share_kv / share_qkv / *_swizzle_qkv entries (round 4: attn_bigd7, either V layout)
timeout 600 $R $S/flash-attn/flash_attn_mma.py --B 1 --H 48 --N 8192 --D 256 --check --show-all --sdpa --seed 1 > $OUT/f1_fa_1x48x8192x256_check.log 2>&1; echo "fa d256 rc=$?" | tee -a $OUT/f1_steps.log
( cd _refstage && sha256sum -c SHA256SUMS ) > $OUT/f1_sha256_after.txt 2>&1
tail -3 $OUT/f1_*.log | tail -60
