python tools/attn_rate.py --seconds 1.0 --rounds 3 1,48,8192,1024 1,48,8192,1024:d1024=4 1,48,8192,1024:d1024=6 1,48,8192,1024:d1024=8 1,8,8192,1024 > gpurun_out/r4e_bigd4_span.log 2>&1
cat gpurun_out/r4e_bigd4_span.log
bash tools/pmc_bigd4.sh r4e > /dev/null 2>&1
cat gpurun_out/r4e/pmc_bigd4.txt
