mkdir -p gpurun_out/r2w
timeout 400 python -m pytest tests/test_gpu_attn.py -m gpu -x -q -k "512 or workgroup_shapes or spike or extreme" > gpurun_out/r2w/pytest.log 2>&1; tail -3 gpurun_out/r2w/pytest.log
tools/power_watch.sh r2w -- bash -c "python tools/sustain.py --seconds 2.0 attn attn:nw=512 attn8k attn8k:nw=512 attn:nw=512:zero attn:zero attn:nw=512 attn" > gpurun_out/r2w/run.log 2>&1
cat gpurun_out/r2w/run.log
