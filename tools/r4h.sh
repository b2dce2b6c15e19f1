python -m pytest tests/test_gpu_attn.py -q -x 2>&1 | tail -8 > gpurun_out/r4h_pytest.log
tail -4 gpurun_out/r4h_pytest.log
python tools/attn_rate.py --seconds 1.0 --rounds 3 1,48,8192,64:lsum=1 1,48,8192,64:lsum=2 1,8,8192,64:lsum=1 1,8,8192,64:lsum=2 4,32,4096,64:lsum=1 4,32,4096,64:lsum=2 1,48,8192,64:vt:lsum=1 1,48,8192,64:vt:lsum=2 1,48,8192,64:zero:lsum=1 1,48,8192,64:zero:lsum=2 > gpurun_out/r4h_attn_lsum.log 2>&1
cat gpurun_out/r4h_attn_lsum.log
