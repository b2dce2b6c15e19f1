#!/bin/bash
# round 6, first GPU call: the reference's default HGEMM sweep against hipBLASLt + which kernels hipBLASLt picks
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r6a; mkdir -p $OUT
python tools/hgemm_sizes.py sweep $OUT/sweep.json 256 12800 256 0.3 > $OUT/sweep.log 2>&1
rocprofv3 --kernel-trace -d $OUT/vk -o vk -- python tools/vendor_kernels.py run 256,512,768,1024,1280,1536,1792,2048,2304,2560,2816,3072,3328,3584,3840,4096,4608,5120,6144 > $OUT/vk_run.log 2>&1
python tools/vendor_kernels.py report $OUT/vk > $OUT/vk_report.log 2>&1
find $OUT -name "*.db" -size +20M -delete
timeout 900 python -m pytest tests/test_gpu_attn.py -q -x -k "config3 or agree_with_each_other" > $OUT/pytest_new.log 2>&1
tail -5 $OUT/pytest_new.log
