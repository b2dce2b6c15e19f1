#!/usr/bin/env python3
"""A/B of the ragged-last-wave split (lc_tune_set "hgemm_tail"): sizes whose 256-tile grid ends in a short wave, interleaved
rounds, >= S seconds sustained per cell, TN and NN, hipBLASLt alongside.  usage: hgemm_tail_ab.py [sizes,comma] [seconds] [rounds]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from leetcuda_amd import capi, host  # noqa: E402

capi.load()
capi.vendor_init()
sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [4608, 6144, 12544, 15360, 15872]
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3


def rate(step, fl):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(5):
        step()
    t1.record(); torch.cuda.synchronize()
    n = max(5, int(secs / (t0.elapsed_time(t1) / 5 * 1e-3)))
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
    T = (n // 256) ** 2
    for lname, lay in (("tn", capi.LAYOUT_TN), ("nn", capi.LAYOUT_NN)):
        b2 = host.as_col_major(b) if lay == capi.LAYOUT_TN else b
        res = {"one": [], "split": [], "vendor": []}
        for r in range(rounds):
            for name, knob in (("one", 0), ("split", 1)):
                capi.tune("hgemm_tail", knob)
                try:
                    res[name].append(rate(lambda: capi.hgemm(a, b2, c, layout=lay, variant=capi.HGEMM_AUTO, swizzle_stride=st), fl))
                finally:
                    capi.tune("hgemm_tail", 1)
            res["vendor"].append(rate(lambda: capi.hgemm_vendor(a, b2, c, lay), fl))
        med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
        print(f"n={n:5d} {lname} ({T} tiles = {T / 256:.2f} waves, remainder {T % 256}): one launch {med['one']:7.1f} | split {med['split']:7.1f} "
              f"({(med['split'] / med['one'] - 1) * 100:+.1f} %) | hipBLASLt {med['vendor']:7.1f}", flush=True)
    del a, b, c
capi.vendor_destroy()
