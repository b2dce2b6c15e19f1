"""fp8 GEMM rate per kernel form (lc_tune_set "fp8_mx") and for the MX entry with real block scales.
usage: python tools/fp8_rate.py [n ...]   (default 16384 8192);  randn e4m3 inputs, >= 1.5 s sustained per row"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from leetcuda_amd import capi


def rate(fn, flops, seconds=1.5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    n = max(3, int(seconds * 1e3 / e0.elapsed_time(e1)))
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return flops * n / (e0.elapsed_time(e1) * 1e-3) * 1e-12


def main():
    capi.load()
    sizes = [int(x) for x in sys.argv[1:]] or [16384, 8192]
    for n in sizes:
        torch.manual_seed(0)
        a = torch.randn(n, n, device="cuda").to(torch.float8_e4m3fn)
        b = torch.randn(n, n, device="cuda").to(torch.float8_e4m3fn)
        c = torch.empty(n, n, dtype=torch.half, device="cuda")
        fl = 2.0 * n ** 3
        for mx, name in ((3, "K=128 generated"), (1, "K=64 4-wave"), (3, "K=128 generated"), (1, "K=64 4-wave")):
            capi.tune("fp8_mx", mx)
            print(f"n={n} fp8_mx={mx} ({name}): {rate(lambda: capi.gemm_fp8(a, b, c, alpha=1 / 16, swizzle_stride=2048), fl):8.1f} TFLOP/s", flush=True)
        capi.tune("fp8_mx", 3)
        for spread in (0, 2):
            sa = torch.randint(127 - spread, 128 + spread, (n, n // 32), device="cuda", dtype=torch.uint8)
            sb = torch.randint(127 - spread, 128 + spread, (n, n // 32), device="cuda", dtype=torch.uint8)
            pa, pb = capi.mxfp8_pack_scales(sa), capi.mxfp8_pack_scales(sb)
            print(f"n={n} lc_gemm_mxfp8, block scales 2^-{spread}..2^{spread}: "
                  f"{rate(lambda: capi.gemm_mxfp8(a, pa, b, pb, c, alpha=1 / 64, swizzle_stride=2048), fl):8.1f} TFLOP/s", flush=True)
        z = torch.zeros_like(a)
        print(f"n={n} zero-filled, K=128: {rate(lambda: capi.gemm_fp8(z, z, c, swizzle_stride=2048), fl):8.1f} TFLOP/s", flush=True)
        capi.tune("fp8_mx", 1)
        print(f"n={n} zero-filled, K=64 : {rate(lambda: capi.gemm_fp8(z, z, c, swizzle_stride=2048), fl):8.1f} TFLOP/s", flush=True)
        capi.tune("fp8_mx", 3)


if __name__ == "__main__":
    main()
