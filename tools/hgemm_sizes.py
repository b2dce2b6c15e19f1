#!/usr/bin/env python3
"""Variant x size table (square fp16 GEMM, TN and NN) incl. hipBLASLt: data for the LC_HGEMM_AUTO heuristic.

    tools/hgemm_sizes.py [sizes,comma,separated] [variants,comma,separated] [seconds]
    tools/hgemm_sizes.py published      the sizes the reference publishes its C++-bench numbers on (kernels/hgemm/README.md:159-185:
                                        12544, 15360, 15616, 15872, 16128, 16384) + 8192, AUTO and hipBLASLt, 0.6 s sustained per cell
    tools/hgemm_sizes.py sweep [out.json] [lo] [hi] [step] [seconds]
                                        the reference bench's DEFAULT sweep (kernels/hgemm/hgemm.py:28-32,419-421: M = N = K, every multiple of
                                        256 up to 12800), AUTO against hipBLASLt, TN + NN, `seconds` (default 0.3) sustained per cell split
                                        into three rounds that alternate the two libraries (one box state for both sides); prints the
                                        ratio per cell, the worst cells, and writes the table as JSON"""
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
SWEEP = len(sys.argv) > 1 and sys.argv[1] == "sweep"
if SWEEP:
    sizes = []
elif len(sys.argv) > 1 and sys.argv[1] == "published":
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


def burst(step, n):
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n):
        step()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) * 1e-3


def sweep():
    import json
    out = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/hgemm_sweep.json"
    lo, hi, stp = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((3, 256), (4, 12800), (5, 256)))
    seconds = float(sys.argv[6]) if len(sys.argv) > 6 else 0.3
    rounds = 3
    cells = []
    for n in range(lo, hi + 1, stp):
        a = torch.randn(n, n, dtype=torch.half, device="cuda")
        b = torch.randn(n, n, dtype=torch.half, device="cuda")
        c = torch.empty(n, n, dtype=torch.half, device="cuda")
        fl = 2.0 * n ** 3
        st = host.make_block_swizzle_stride(n, n)
        for lname, lay in (("tn", capi.LAYOUT_TN), ("nn", capi.LAYOUT_NN)):
            b2 = host.as_col_major(b) if lay == capi.LAYOUT_TN else b
            ours = lambda: capi.hgemm(a, b2, c, layout=lay, variant=capi.HGEMM_AUTO, swizzle_stride=st)   # noqa: E731
            vend = lambda: capi.hgemm_vendor(a, b2, c, lay)   # noqa: E731
            for f in (ours, vend):
                burst(f, 3)
            per = max(burst(ours, 10), burst(vend, 10)) / 10
            cnt = max(10, int(seconds / rounds / per))
            t = [0.0, 0.0]
            for r in range(rounds):
                for i in ((0, 1) if r % 2 == 0 else (1, 0)):
                    t[i] += burst(ours if i == 0 else vend, cnt)
            to, tv = (fl * cnt * rounds / x * 1e-12 for x in t)
            cells.append({"n": n, "layout": lname, "auto_tflops": round(to, 1), "hipblaslt_tflops": round(tv, 1), "ratio": round(to / tv, 4),
                          "kernel": capi.hgemm_kernel_name(n, n, n, lay) if hasattr(capi, "hgemm_kernel_name") else None,
                          "launches_per_side": cnt * rounds})
            print(f"n={n:5d} {lname}: auto {to:7.1f} | hipBLASLt {tv:7.1f} | ratio {to / tv:6.3f} | {cells[-1]['kernel']}", flush=True)
        del a, b, c
    worst = sorted(cells, key=lambda x: x["ratio"])[:12]
    print("worst cells:", [(w["n"], w["layout"], w["ratio"]) for w in worst])
    for lname in ("tn", "nn"):
        rs = [x["ratio"] for x in cells if x["layout"] == lname]
        print(f"{lname}: min {min(rs):.3f} geomean {torch.tensor(rs).log().mean().exp().item():.3f} cells < 0.97: {sum(r < 0.97 for r in rs)} of {len(rs)}")
    Path(out).parent.mkdir(parents=True, exist_ok=True)
    Path(out).write_text(json.dumps({"seconds_per_cell_per_side": seconds, "rounds_interleaved": rounds, "device": torch.cuda.get_device_name(0),
                                     "cells": cells}, indent=1))


if SWEEP:
    sweep()
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
