#!/usr/bin/env python3
"""Fuzz of LC_HGEMM_AUTO on shapes no tile divides (late round 6: LC_HGEMM_RAGGED, LC_HGEMM_KPAD, hgemm_mid_edge_kernel and its split-K, hgemm_edge_kernel):
random (M, N, K, layout) with N % 8 == 0 and K % 8 == 0 — a third of them with K % 32 == 0 — in four size classes (tiny, one-round, multi-round, flagship-sized
interior), whatever the dispatcher picks against hgemm_generic_kernel (element-wise staging: the independent kernel) to one output ulp, NaN canaries around C,
a second run bit for bit.  Prints the kernel families seen; exits 1 on the first mismatch.

    tools/hgemm_ragged_fuzz.py [cases] [seed]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from leetcuda_amd import capi, host  # noqa: E402

capi.load()
CASES = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
seen = {}
for i in range(CASES):
    cls = i % 4
    hi = (300, 1500, 3500, 6000)[cls]
    lo = (1, 100, 1500, 4200)[cls]
    M = int(rng.integers(lo, hi))
    N = int(rng.integers(max(1, lo // 8), hi // 8 + 1)) * 8
    kq = int(rng.integers(1, (64, 520, 300, 80)[cls]))
    K = kq * 32 if rng.random() < 0.4 else kq * 32 + int(rng.integers(1, 4)) * 8
    lay = capi.LAYOUT_NN if rng.random() < 0.5 else capi.LAYOUT_TN
    torch.manual_seed(i)
    a = torch.randn(M, K, dtype=torch.half, device="cuda")
    b = torch.randn(K, N, dtype=torch.half, device="cuda")
    bb = host.as_col_major(b) if lay == capi.LAYOUT_TN else b
    name = capi.hgemm_kernel_name(M, N, K, lay)
    fam = " + ".join(p.split("<")[0] for p in name.split(" + "))
    seen[fam] = seen.get(fam, 0) + 1
    pad = 4096
    buf = torch.full((M * N + 2 * pad,), float("nan"), dtype=torch.half, device="cuda")
    c = buf[pad:pad + M * N].view(M, N)
    stride = host.make_block_swizzle_stride(N, K)
    capi.hgemm(a, bb, c, layout=lay, variant=capi.HGEMM_AUTO, swizzle_stride=stride)
    c2 = torch.empty(M, N, dtype=torch.half, device="cuda")
    capi.hgemm(a, bb, c2, layout=lay, variant=capi.HGEMM_AUTO, swizzle_stride=stride)
    g = torch.empty(M, N, dtype=torch.half, device="cuda")
    capi.hgemm(a, bb, g, layout=lay, variant=capi.HGEMM_GENERIC)
    torch.cuda.synchronize()
    ulp = torch.clamp(g.float().abs(), min=32.0) * 2.0 ** -10
    bad = []
    if not (torch.isnan(buf[:pad]).all() and torch.isnan(buf[pad + M * N:]).all()):
        bad.append("wrote outside C")
    if not torch.isfinite(c).all():
        bad.append("non-finite output")
    elif not ((c.float() - g.float()).abs() <= ulp).all():
        d = (c.float() - g.float()).abs() - ulp
        idx = int(d.argmax())
        bad.append(f"differs from hgemm_generic_kernel by more than one ulp at ({idx // N}, {idx % N}): {c.flatten()[idx].item()} vs {g.flatten()[idx].item()}")
    if not torch.equal(c, c2):
        bad.append("second run differs")
    if bad:
        print(f"FAIL case {i}: {M}x{N}x{K} {'nn' if lay == capi.LAYOUT_NN else 'tn'} {name}: {'; '.join(bad)}", flush=True)
        sys.exit(1)
    if i % 20 == 0:
        print(f"case {i}: {M}x{N}x{K} {'nn' if lay == capi.LAYOUT_NN else 'tn'} {name} ok", flush=True)
print(f"{CASES} cases ok; kernel families: " + ", ".join(f"{k} x{v}" for k, v in sorted(seen.items(), key=lambda kv: -kv[1])))
