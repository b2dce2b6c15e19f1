mkdir -p gpurun_out/r2u
tools/power_watch.sh r2u -- bash -c "python tools/sustain.py --seconds 2.0 attn attn:zero attn8k attn8k:zero d512 d512:zero attn:nw=64 attn:nw=8" > gpurun_out/r2u/run.log 2>&1
cat gpurun_out/r2u/run.log
timeout 300 python bench.py --no-cpu-baseline --no-attention > gpurun_out/r2u/bench.json 2> gpurun_out/r2u/bench.err; python -c "
import json; d=json.load(open('gpurun_out/r2u/bench.json')); print({k:d[k] for k in ('value','vendor_tflops','sustained','uniform_tflops')})"
