import sys, time
sys.path.insert(0, ".")
import torch
from leetcuda_amd import capi
capi.load()
for nn in (16384, 8192):
    a8 = torch.randn(nn, nn, device="cuda").to(torch.float8_e4m3fn)
    b8 = torch.randn(nn, nn, device="cuda").to(torch.float8_e4m3fn)
    c8 = torch.zeros(nn, nn, dtype=torch.half, device="cuda")
    for rnd in range(3):
        for name, knob in (("stagger off", 1 << 27), ("stagger default", 0)):
            capi.tune("hgemm_stagger", knob)
            for _ in range(3):
                capi.gemm_fp8(a8, b8, c8, alpha=1 / 16, swizzle_stride=2048)
            torch.cuda.synchronize()
            n = 400 if nn == 16384 else 2500
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                capi.gemm_fp8(a8, b8, c8, alpha=1 / 16, swizzle_stride=2048)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            print(f"fp8 {nn}^3 {name:16s}: {ms:.4f} ms {2.0 * nn ** 3 / ms * 1e-9:8.1f} TFLOP/s", flush=True)
    capi.tune("hgemm_stagger", 0)
    del a8, b8, c8
