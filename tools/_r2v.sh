mkdir -p gpurun_out/r2v
timeout 400 python -m pytest tests/test_gpu_attn.py -m gpu -x -q -k "512 or workgroup_shapes or spike or extreme" > gpurun_out/r2v/pytest.log 2>&1; tail -8 gpurun_out/r2v/pytest.log
timeout 200 python tools/attn_ab.py 2 > gpurun_out/r2v/ab.log 2>&1; grep -v amdgpu gpurun_out/r2v/ab.log | tail -22
