python -m pytest tests/test_gpu_attn.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py -q -k "full_width or d512 or config5a or bf16_large" 2>&1 | tail -8 > gpurun_out/r4k_pytest.log
tail -4 gpurun_out/r4k_pytest.log
python tools/attn_rate.py --seconds 1.0 --rounds 3 1,48,8192,512 1,48,8192,512:d512=3 1,48,8192,512:bf16 1,48,8192,512:bf16:d512=3 1,48,8192,512:zero 1,48,8192,512:zero:d512=3 > gpurun_out/r4k_bigd6.log 2>&1
cat gpurun_out/r4k_bigd6.log
