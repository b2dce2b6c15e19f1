#!/usr/bin/env python3
"""Sustained rate of the fp16 GEMM at arbitrary (M, N, K) — the reference's legal shapes are multiples of 128 x 128 x 32
(hgemm_mma_stage.cu:650,675-676), not only cubes of 256 — per variant and layout, next to hipBLASLt.
usage: hgemm_shapes.py [--seconds S] M,N,K[:auto|w4y|mfma128|generic|vendor ...]
Every cell is >= S seconds of back-to-back launches (both sides sit at the board's power cap from ~4096^3 on); prints the kernel the
dispatcher reports, TFLOP/s and the ratio to the first spec's same-layout rate (normally 8192,8192,8192 as the yardstick)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from leetcuda_amd import capi, host  # noqa: E402

args = sys.argv[1:]
secs = 0.6
if args and args[0] == "--seconds":
    secs = float(args[1])
    args = args[2:]
capi.load()
capi.vendor_init()
V = {"auto": capi.HGEMM_AUTO, "w4y": capi.HGEMM_MFMA256W4Y, "mfma128": capi.HGEMM_MFMA128, "generic": capi.HGEMM_GENERIC}


def rate(step, fl):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(5):
        step()
    t1.record()
    torch.cuda.synchronize()
    n = max(5, int(secs / (t0.elapsed_time(t1) / 5 * 1e-3)))
    t0.record()
    for _ in range(n):
        step()
    t1.record()
    torch.cuda.synchronize()
    return fl / (t0.elapsed_time(t1) / n) * 1e-9


yard = {}
for spec in args:
    shape, *vs = spec.split(":")
    M, N, K = (int(x) for x in shape.split(","))
    vs = vs or ["auto", "vendor"]
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, dtype=torch.half, device="cuda")
    b = torch.randn(K, N, dtype=torch.half, device="cuda")
    c = torch.empty(M, N, dtype=torch.half, device="cuda")
    fl = 2.0 * M * N * K
    st = host.make_block_swizzle_stride(N, K)
    for lname, lay in (("tn", capi.LAYOUT_TN), ("nn", capi.LAYOUT_NN)):
        b2 = host.as_col_major(b) if lay == capi.LAYOUT_TN else b
        for v in vs:
            if v == "vendor":
                tf, name = rate(lambda: capi.hgemm_vendor(a, b2, c, lay), fl), "hipBLASLt"
            else:
                try:
                    name = capi.hgemm_kernel_name(M, N, K, lay, V[v])
                except capi.LcError as e:
                    print(f"SHAPE {shape:18s} {lname} {v:8s} refused: {e}", flush=True)
                    continue
                tf = rate(lambda: capi.hgemm(a, b2, c, layout=lay, variant=V[v], swizzle_stride=st), fl)
            y = yard.setdefault((lname, v), tf)
            print(f"SHAPE {shape:18s} {lname} {v:8s} {name:28s} {tf:7.1f} TFLOP/s  ({tf / y:.3f} of the first spec's)", flush=True)
    del a, b, c
capi.vendor_destroy()
