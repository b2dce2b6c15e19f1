#!/bin/bash
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r6g; mkdir -p $OUT
python tools/attn_mix_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/attn_mix_probe.log
python tools/hgemm_stagger_ab.py 4096,4864,7680,8192,8960,10240,12288 0.6 tn 2>&1 | grep -v amdgpu.ids | tee $OUT/stagger_ab.log
