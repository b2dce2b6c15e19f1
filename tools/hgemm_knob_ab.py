#!/usr/bin/env python3
"""A/B of launch knobs of LC_HGEMM_AUTO on large shapes against hipBLASLt (sustained, candidates rotated over three rounds).
    tools/hgemm_knob_ab.py [sizes] [seconds] [tn|nn|both]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from leetcuda_amd import capi, host  # noqa: E402

capi.load()
capi.vendor_init()
sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [8960, 12288]
SECONDS = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
LAYS = sys.argv[3] if len(sys.argv) > 3 else "tn"


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
    st0 = host.make_block_swizzle_stride(n, n)
    for lname, lay in (("tn", capi.LAYOUT_TN), ("nn", capi.LAYOUT_NN)):
        if LAYS not in ("both", lname):
            continue
        b2 = host.as_col_major(b) if lay == capi.LAYOUT_TN else b
        if len(sys.argv) > 4 and sys.argv[4] == "tail":
            sets = {"auto": {}, "eighths": {"hgemm_tail_tile": 1}, "quadrants": {"hgemm_tail_tile": 2}, "tail2 (mfma128 + split-K)": {"hgemm_tail": 2}, "tail0 (one launch)": {"hgemm_tail": 0}, "tail3 (R <= 0.75 CUs)": {"hgemm_tail": 3},
                    "tail4 (any R)": {"hgemm_tail": 4}}
        elif len(sys.argv) > 4 and sys.argv[4] == "sched":
            sets = {"auto": {}, "sched1": {"w4y_sched": 1}, "sched2": {"w4y_sched": 2}, "raster1": {"hgemm_raster": 1}, "raster2": {"hgemm_raster": 2},
                    "raster2+sched2": {"hgemm_raster": 2, "w4y_sched": 2}, "raster1+sched2": {"hgemm_raster": 1, "w4y_sched": 2}}
        else:
            sets = {"auto": {}, "raster1": {"hgemm_raster": 1}, "raster2": {"hgemm_raster": 2}, "persist0": {"hgemm_persist": 0}, "tail0": {"hgemm_tail": 0},
                    "raster1 s1024": {"hgemm_raster": 1, "_stride": 1024}, "raster1 s2048": {"hgemm_raster": 1, "_stride": 2048},
                    "raster1 s4096": {"hgemm_raster": 1, "_stride": 4096}, "sched0": {"w4y_sched": 0}, "sched2": {"w4y_sched": 2}}

        def mk(kn):
            def f():
                for k, v in kn.items():
                    if not k.startswith("_"):
                        capi.tune(k, v)
                capi.hgemm(a, b2, c, layout=lay, variant=capi.HGEMM_AUTO, swizzle_stride=kn.get("_stride", st0))
                for k in kn:
                    if not k.startswith("_"):
                        capi.tune(k, capi.tune_get(k)[1])
            return f
        cands = {k: mk(v) for k, v in sets.items()}
        cands["hipBLASLt"] = lambda: capi.hgemm_vendor(a, b2, c, lay)
        for f in cands.values():
            burst(f, 2)
        per = burst(cands["auto"], 5) / 5
        cnt = max(5, int(SECONDS / 3 / per))
        t = {k: 0.0 for k in cands}
        keys = list(cands)
        for r in range(3):
            for k in keys[r:] + keys[:r]:
                t[k] += burst(cands[k], cnt)
        rate = {k: fl * cnt * 3 / v * 1e-12 for k, v in t.items()}
        print(f"n={n:5d} {lname} (stride {st0}): " + " | ".join(f"{k} {v:6.1f}" for k, v in rate.items()), flush=True)
capi.vendor_destroy()
