set -x
python -m pytest tests/test_gpu_attn.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py -q -x -k "not hgemm" 2>&1 | tail -15 > gpurun_out/r4c_pytest.log
tail -5 gpurun_out/r4c_pytest.log
python tools/attn_rate.py --seconds 1.0 --rounds 3 4,32,4096,128:nw=513 4,32,4096,128:nw=515 4,32,4096,128:nw=517 4,32,4096,128:vt:nw=515 4,32,4096,128:vt:nw=517 4,32,8192,128:nw=513 4,32,8192,128:nw=515 4,32,8192,128:nw=517 4,32,8192,128:vt:nw=517 1,48,8192,64:nw=513 1,48,8192,64:nw=515 1,48,8192,64:nw=517 1,48,8192,64:vt:nw=517 4,32,2048,128:nw=513 4,32,2048,128:nw=515 4,32,2048,128:nw=517 > gpurun_out/r4c_attn_walks.log 2>&1
cat gpurun_out/r4c_attn_walks.log
