#!/usr/bin/env python3
"""A/B of the mid-size HGEMM kernel (hgemm_mid.hip) per size: every legal (tile, ring depth) against the launch LC_HGEMM_AUTO made
before it existed ("hgemm_mid" = 1) and against hipBLASLt; `seconds` sustained per cell in three rounds that rotate the candidates.

    tools/hgemm_mid_ab.py [sizes,comma] [seconds] [tn|nn|both]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from leetcuda_amd import capi, host  # noqa: E402

capi.load()
capi.vendor_init()
sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1280, 1536, 1792, 2048, 2304, 2560, 2816, 3072]
SECONDS = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
LAYS = sys.argv[3] if len(sys.argv) > 3 else "both"


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
    for lname, lay in (("tn", capi.LAYOUT_TN), ("nn", capi.LAYOUT_NN)):
        if LAYS not in ("both", lname):
            continue
        b2 = host.as_col_major(b) if lay == capi.LAYOUT_TN else b
        cands = {}

        def mk(mid, ns, var):
            def f():
                capi.tune("hgemm_mid", mid)
                capi.tune("hgemm_mid_ns", ns)
                capi.hgemm(a, b2, c, layout=lay, variant=var, swizzle_stride=st)
            return f
        cands["old-auto"] = mk(1, 0, capi.HGEMM_AUTO)
        cands["auto"] = mk(0, 0, capi.HGEMM_AUTO)
        for tmw in (2, 3, 1):
            for w in (2, 3):
                if n % (64 * w) or n % (64 * tmw) or (lname == "nn" and w == 3) or (tmw == 3 and w == 3):
                    continue
                for ns in (2, 3):
                    cands[f"mid{tmw}{w}x{ns}"] = mk(10 * tmw + w, ns, capi.HGEMM_MID)
        cands["hipBLASLt"] = lambda: capi.hgemm_vendor(a, b2, c, lay)
        for f in cands.values():
            burst(f, 3)
        per = burst(cands["old-auto"], 10) / 10
        cnt = max(10, int(SECONDS / 3 / per))
        t = {k: 0.0 for k in cands}
        keys = list(cands)
        for r in range(3):
            for k in keys[r:] + keys[:r]:
                t[k] += burst(cands[k], cnt)
        capi.tune("hgemm_mid", 0)
        capi.tune("hgemm_mid_ns", 0)
        rate = {k: fl * cnt * 3 / v * 1e-12 for k, v in t.items()}
        best = max((k for k in rate if k.startswith("mid")), key=lambda k: rate[k], default=None)
        print(f"n={n:5d} {lname} auto={capi.hgemm_kernel_name(n, n, n, lay)}: " + " | ".join(f"{k} {v:6.1f}" for k, v in rate.items())
              + (f" || best {best} {rate[best] / rate['hipBLASLt']:.3f} of hipBLASLt, {rate[best] / rate['old-auto']:.3f} of old" if best else ""), flush=True)
capi.vendor_destroy()
