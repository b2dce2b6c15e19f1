#!/usr/bin/env python3
"""fp8 e4m3 GEMM (BASELINE config 5) timing: MX-scaled K=64 MFMA with unit scales vs the plain K=16 fp8 MFMA."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from leetcuda_amd import capi  # noqa: E402
capi.load()
for nn in (8192, 16384):
    a8 = torch.randn(nn, nn, device="cuda").to(torch.float8_e4m3fn)
    b8 = torch.randn(nn, nn, device="cuda").to(torch.float8_e4m3fn)
    c8 = torch.zeros(nn, nn, dtype=torch.half, device="cuda")
    for mx in (1, 2, 0, 1, 2, 0):
        capi.tune("fp8_mx", mx)
        for st in (2048,):
            for _ in range(3):
                capi.gemm_fp8(a8, b8, c8, alpha=1 / 16, swizzle_stride=st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                capi.gemm_fp8(a8, b8, c8, alpha=1 / 16, swizzle_stride=st)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            print(f"fp8 {nn}^3 {('plain k16   ', 'mx k64 4-wave', 'mx k64 8-wave')[mx]} stride {st}: {ms:.4f} ms {2.0 * nn ** 3 / ms * 1e-9:8.1f} TFLOP/s", flush=True)
    del a8, b8, c8
capi.tune("fp8_mx", 1)
