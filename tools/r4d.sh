python -m pytest tests/test_gpu_attn.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py -q -k "not hgemm" 2>&1 | tail -25 > gpurun_out/r4d_pytest.log
tail -8 gpurun_out/r4d_pytest.log
python tools/attn_rate.py --seconds 1.0 --rounds 3 1,48,8192,1024 1,48,8192,1024:d512=1 1,48,8192,256 1,48,8192,256:vt 1,48,8192,512 1,48,4096,1024 > gpurun_out/r4d_attn_bigd.log 2>&1
cat gpurun_out/r4d_attn_bigd.log
