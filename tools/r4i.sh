python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_attn.py -q -k "1024 or large_head" 2>&1 | tail -6 > gpurun_out/r4i_pytest.log
tail -3 gpurun_out/r4i_pytest.log
python tools/attn_rate.py --seconds 1.0 --rounds 3 1,48,8192,1024 1,48,8192,1024:d1024=4 1,48,8192,1024:d1024=6 1,48,8192,1024:d1024=8 1,48,4096,1024 1,48,8192,1024:zero > gpurun_out/r4i_bigd4_v2.log 2>&1
cat gpurun_out/r4i_bigd4_v2.log
