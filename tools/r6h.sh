#!/bin/bash
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r6h; mkdir -p $OUT
python tools/attn_mix_probe.py 2>&1 | grep -v amdgpu.ids | head -9 | tee $OUT/attn_mix_probe.log
python tools/hgemm_knob_ab.py 4096,4864,8192,8960,12288 0.6 tn 2>&1 | grep -v amdgpu.ids | tee $OUT/knob_ab.log
rocprofv3 --kernel-trace -d $OUT/vk -o vk -- python tools/vendor_kernels.py run 4864,7680,8192,8960,10240,12288 tn > $OUT/vk_run.log 2>&1
python tools/vendor_kernels.py report $OUT/vk 2>&1 | tee $OUT/vk_report.log | cut -c1-260
find $OUT -name "*.db" -delete
timeout 900 python -m pytest tests/test_gpu_attn.py -q -x -k "split or calibration" > $OUT/pytest_split.log 2>&1; tail -4 $OUT/pytest_split.log
