python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r4f_pytest.log
tail -6 gpurun_out/r4f_pytest.log
python tools/attn_rate.py --seconds 1.0 --rounds 3 1,48,8192,1024 1,48,8192,1024:d1024=2 1,48,8192,1024:d1024=6 > gpurun_out/r4f_bigd4.log 2>&1
cat gpurun_out/r4f_bigd4.log
