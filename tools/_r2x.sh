mkdir -p gpurun_out/r2x
timeout 500 python -m pytest tests/test_gpu_hgemm.py -m gpu -x -q > gpurun_out/r2x/pytest.log 2>&1; tail -5 gpurun_out/r2x/pytest.log
python - <<'PY' 2>&1 | grep -v amdgpu
import torch, sys
sys.path.insert(0, '.')
from leetcuda_amd import capi
capi.load()
n = 4096
a = torch.randn(n, n, dtype=torch.half, device="cuda"); b = torch.randn(n, n, dtype=torch.half, device="cuda"); c = torch.empty(n, n, dtype=torch.half, device="cuda")
for v in range(20, 31):
    it = 2 if v < 22 else 10
    ms = capi.hgemm_time(a, b, c, capi.LAYOUT_NN, v, 2, 1, warmup=1, iters=it)
    print(f"rung {v} {capi.hgemm_kernel_name(n, n, n, capi.LAYOUT_NN, v):52s} {ms:9.3f} ms {2 * n ** 3 / ms * 1e-9:8.2f} TFLOP/s", flush=True)
PY
