#!/usr/bin/env python3
"""Variant x size table (square fp16 GEMM, TN and NN) incl. hipBLASLt: data for the LC_HGEMM_AUTO heuristic.

    tools/hgemm_sizes.py [sizes,comma,separated] [variants,comma,separated] [seconds]
    tools/hgemm_sizes.py published      the sizes the reference publishes its C++-bench numbers on (kernels/hgemm/README.md:159-185:
                                        12544, 15360, 15616, 15872, 16128, 16384) + 8192, AUTO and hipBLASLt, 0.6 s sustained per cell"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from leetcuda_amd import capi, host  # noqa: E402

capi.load()
capi.vendor_init()
V = {"mfma128": capi.HGEMM_MFMA128, "pingpong2": capi.HGEMM_MFMA256P2, "w4c": capi.HGEMM_MFMA256W4C,
     "w4y": capi.HGEMM_MFMA256W4Y, "auto": capi.HGEMM_AUTO}
SECONDS = 0.3
if len(sys.argv) > 1 and sys.argv[1] == "published":
    sizes = [8192, 12544, 15360, 15616, 15872, 16128, 16384]
    V = {"auto": capi.HGEMM_AUTO}
    SECONDS = 0.6
else:
    sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1024, 2048, 3072, 4096, 6144, 8192]
    if len(sys.argv) > 2:
        V = {k: V[k] for k in sys.argv[2].split(",")}
    if len(sys.argv) > 3:
        SECONDS = float(sys.argv[3])


def rate(step, fl, seconds=None):
    seconds = SECONDS if seconds is None else seconds
    """>= `seconds` of back-to-back launches (both sides run at the power cap from ~4096^3 on: short bursts mislead)"""
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(10):
        step()
    t1.record(); torch.cuda.synchronize()
    n = max(10, int(seconds / (t0.elapsed_time(t1) / 10 * 1e-3)))
    t0.record()
    for _ in range(n):
        step()
    t1.record(); torch.cuda.synchronize()
    return fl / (t0.elapsed_time(t1) / n) * 1e-9
for n in sizes:
    a = torch.randn(n, n, dtype=torch.half, device="cuda")
    b = torch.randn(n, n, dtype=torch.half, device="cuda")
    c = torch.empty(n, n, dtype=torch.half, device="cuda")
    fl = 2.0 * n ** 3
    st = host.make_block_swizzle_stride(n, n)
    for lname, lay in (("tn", capi.LAYOUT_TN), ("nn", capi.LAYOUT_NN)):
        b2 = host.as_col_major(b) if lay == capi.LAYOUT_TN else b
        row = []
        for name, var in V.items():
            if name != "mfma128" and name != "auto" and n % 256:
                continue
            row.append(f"{name} {rate(lambda: capi.hgemm(a, b2, c, layout=lay, variant=var, swizzle_stride=st), fl):7.1f}")
        row.append(f"hipBLASLt {rate(lambda: capi.hgemm_vendor(a, b2, c, lay), fl):7.1f}")
        print(f"n={n:5d} {lname} (block-swizzle stride {st}): " + " | ".join(row), flush=True)
capi.vendor_destroy()
