python -m pytest tests/test_gpu_hgemm.py tests/test_gpu_fullsize.py -q -k "hgemm or persistent or concurrent or tuned or ragged or raster or stagger or full_size_config2" 2>&1 | tail -8 > gpurun_out/r4g_pytest.log
tail -4 gpurun_out/r4g_pytest.log
python tools/sustain.py --seconds 1.5 hgemm hgemm:nopersist vendor hgemm hgemm:nopersist vendor hgemm hgemm:nopersist vendor hgemm:nn hgemm:nn:nopersist hgemm:zero hgemm:zero:nopersist vendor:zero > gpurun_out/r4g_hgemm_prefetch.log 2>&1
cat gpurun_out/r4g_hgemm_prefetch.log
