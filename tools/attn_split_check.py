#!/usr/bin/env python3
"""Does the split-KV rule pick the measured-best factor?  (round-5 verdict, next #7.)  For each shape: TFLOP/s with the factor forced to
1 / 2 / 4 / 8 / 16 ("attn_split"), with the rule on its built-in constants ("attn_calib" = 1) and with the constants lc_tune_calibrate
measured on this device; >= 0.25 s sustained per cell, cells rotated over three rounds.  Verdict per shape: the rule's pick against the best
forced factor (ok = within 3 %).      tools/attn_split_check.py [seconds]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from leetcuda_amd import capi  # noqa: E402

capi.load()
SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
cal = capi.tune_calibrate()
print("calibration:", cal, "CUs:", capi.device_check(), flush=True)
SHAPES = [(1, 8, 1024, 128), (1, 8, 2048, 64), (1, 16, 2048, 128), (1, 4, 4096, 128), (1, 2, 8192, 128), (1, 10, 8192, 128), (1, 12, 8192, 64),
          (1, 32, 1024, 128), (1, 6, 8192, 128), (1, 48, 8192, 64), (1, 8, 8192, 64), (2, 8, 4096, 128), (1, 16, 1024, 64), (1, 24, 2048, 128),
          (4, 8, 1024, 128), (1, 3, 4096, 64)]


def burst(step, n):
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n):
        step()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) * 1e-3


good = 0
for (B, H, N, D) in SHAPES:
    q = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    k = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    v = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    o = torch.empty_like(q)
    fl = 4.0 * B * H * N * N * D

    def mk(split, calib):
        def f():
            capi.tune("attn_split", split)
            capi.tune("attn_calib", calib)
            capi.attn_fwd(q, k, v, o)
        return f
    cands = {f"S{s}": mk(s, 0) for s in (1, 2, 4, 8, 16) if s == 1 or ((N // 64) % s == 0 and (N // 64) // s >= 2)}
    cands["rule(built-in)"] = mk(0, 1)
    cands["rule(calibrated)"] = mk(0, 0)
    for f in cands.values():
        burst(f, 3)
    per = burst(cands["S1"], 10) / 10
    cnt = max(10, int(SECONDS / 3 / per))
    t = {kk: 0.0 for kk in cands}
    keys = list(cands)
    for r in range(3):
        for kk in keys[r:] + keys[:r]:
            t[kk] += burst(cands[kk], cnt)
    capi.tune("attn_split", 0)
    capi.tune("attn_calib", 0)
    names = {}
    for calib in (1, 0):
        capi.tune("attn_calib", calib)
        names[calib] = capi.attn_kernel_name(N, D, bh=B * H)
    capi.tune("attn_calib", 0)
    rate = {kk: fl * cnt * 3 / vv * 1e-12 for kk, vv in t.items()}
    best = max((kk for kk in rate if kk.startswith("S")), key=lambda kk: rate[kk])
    ok = rate["rule(calibrated)"] >= 0.97 * rate[best]
    good += ok
    print(f"({B},{H},{N},{D}): " + " | ".join(f"{kk} {vv:6.1f}" for kk, vv in rate.items())
          + f" || best {best}; calibrated rule {rate['rule(calibrated)'] / rate[best]:.3f} of it ({'ok' if ok else 'MISS'}), built-in {rate['rule(built-in)'] / rate[best]:.3f}"
          + f" [{names[0].split('<')[0][9:]}{names[0][names[0].index('<'):]}]", flush=True)
print(f"rule within 3 % of the best forced factor on {good} of {len(SHAPES)} shapes")
