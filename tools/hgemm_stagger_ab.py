#!/usr/bin/env python3
"""A/B of K-loop stagger settings of hgemm_w4y_kernel ("hgemm_stagger" = cx | cm << 4 | cn << 8 | step << 12 | mask << 20: the workgroup
starts its K walk at tile ((cx * XCD + cm * tile_row + cn * tile_col) & mask) * step mod KT) against hipBLASLt, TN (and NN), sustained,
candidates rotated over three rounds.     tools/hgemm_stagger_ab.py [sizes] [seconds] [tn|nn|both]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from leetcuda_amd import capi, host  # noqa: E402

capi.load()
capi.vendor_init()
sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [4096, 8192, 12288]
SECONDS = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
LAYS = sys.argv[3] if len(sys.argv) > 3 else "tn"
OFF = 1 << 27


def enc(cx, cm, cn, step, mask):
    return cx | cm << 4 | cn << 8 | min(max(step, 1), 255) << 12 | mask << 20


def burst(step, n):
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n):
        step()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) * 1e-3


for n in sizes:
    a = torch.randn(n, n, dtype=torch.half, device="cuda")
    b = torch.randn(n, n, dtype=torch.half, device="cuda")
    c = torch.empty(n, n, dtype=torch.half, device="cuda")
    fl = 2.0 * n ** 3
    st = host.make_block_swizzle_stride(n, n)
    kt = n // 64
    for lname, lay in (("tn", capi.LAYOUT_TN), ("nn", capi.LAYOUT_NN)):
        if LAYS not in ("both", lname):
            continue
        b2 = host.as_col_major(b) if lay == capi.LAYOUT_TN else b
        if len(sys.argv) > 4 and sys.argv[4] == "steps":     # odd / neighbouring steps of the by-XCD stagger (auto: KT / 8)
            base = kt // 8
            knobs = {"auto": 0, **{f"x8 step {st_}": enc(1, 0, 0, st_, 7) for st_ in (base - 3, base - 1, base + 1, base + 3, base + 5, base // 2, base // 2 + 1, 2 * base + 1)}}
        else:
            knobs = {"auto": 0, "off": OFF,
                 "x8": enc(1, 0, 0, kt // 8, 7), "x8+m": enc(1, 1, 0, kt // 8, 7), "x8+n": enc(1, 0, 1, kt // 8, 7), "x8+m+n": enc(1, 1, 1, kt // 8, 7),
                 "x+8m/64": enc(1, 8, 0, kt // 64, 63), "x+8n/64": enc(1, 0, 8, kt // 64, 63), "x+8m/32": enc(1, 8, 0, kt // 32, 31),
                     "m+n/16": enc(0, 1, 1, kt // 16, 15), "x/8 s1": enc(1, 0, 0, 1, 7), "x+8m s1": enc(1, 8, 0, 1, 63), "x+8m+n s3": enc(1, 8, 1, 3, 127)}

        def mk(v):
            def f():
                capi.tune("hgemm_stagger", v)
                capi.hgemm(a, b2, c, layout=lay, variant=capi.HGEMM_MFMA256W4Y, swizzle_stride=st)
            return f
        cands = {k: mk(v) for k, v in knobs.items()}
        cands["hipBLASLt"] = lambda: capi.hgemm_vendor(a, b2, c, lay)
        for f in cands.values():
            burst(f, 2)
        per = burst(cands["auto"], 5) / 5
        cnt = max(5, int(SECONDS / 3 / per))
        t = {k: 0.0 for k in cands}
        keys = list(cands)
        for r in range(3):
            for k in keys[r::1] + keys[:r]:
                t[k] += burst(cands[k], cnt)
        capi.tune("hgemm_stagger", 0)
        rate = {k: fl * cnt * 3 / v * 1e-12 for k, v in t.items()}
        print(f"n={n:5d} {lname}: " + " | ".join(f"{k} {v:6.1f}" for k, v in rate.items()), flush=True)
capi.vendor_destroy()
