#!/usr/bin/env python3
"""Sustained rate of the attention forward at arbitrary shapes, interleaved A/B over tuning knobs.
usage: attn_rate.py [--seconds S] [--rounds R] spec...
   spec = B,H,N,D[:bf16][:zero][:vt][:nw=K][:walk=K][:split=K][:d512=K][:sched=K]   (vt = V handed over as [B,H,D,N]; knobs = lc_tune_set keys attn_nw / attn_walk / attn_split / attn_d512 / attn_w4i_sched)
Every spec runs >= S seconds of back-to-back launches per round; R rounds interleave the specs (within-probe A/B,
cdna_hip_programming.md rule 24); prints the kernel name the dispatcher reports, median and best TFLOP/s (matmul FLOPs)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from leetcuda_amd import capi, host  # noqa: E402

args = sys.argv[1:]
secs, rounds = 1.0, 3
while args and args[0].startswith("--"):
    if args[0] == "--seconds":
        secs = float(args[1])
    elif args[0] == "--rounds":
        rounds = int(args[1])
    else:
        raise SystemExit(f"unknown option {args[0]}")
    args = args[2:]
import os  # noqa: E402
if os.environ.get("LC_AB_LIB"):    # A/B of two builds on one box: point the ctypes view at another copy of the library
    capi.LIB_PATH = Path(os.environ["LC_AB_LIB"]).resolve()
capi.load()
KNOBS = {"bigd_stagger": "attn_bigd_stagger", "bigd_map": "attn_bigd_map", "nw": "attn_nw", "walk": "attn_walk", "d512": "attn_d512", "d1024": "attn_d1024", "sched": "attn_w4i_sched", "split": "attn_split"}
cache = {}


def tensors(B, H, N, D, bf16, zero):
    key = (B, H, N, D, bf16, zero)
    if key not in cache:
        cache.clear()                                   # one shape resident at a time
        torch.manual_seed(0)
        dt = torch.bfloat16 if bf16 else torch.half
        mk = (lambda: torch.zeros(B, H, N, D, device="cuda", dtype=dt)) if zero else \
             (lambda: torch.randn(B, H, N, D, device="cuda").to(dt))
        cache[key] = (mk(), mk(), mk(), torch.zeros(B, H, N, D, device="cuda", dtype=dt))
    return cache[key]


def run(spec):
    shape, *opts = spec.split(":")
    B, H, N, D = (int(x) for x in shape.split(","))
    bf16, zero, vt = "bf16" in opts, "zero" in opts, "vt" in opts
    knobs = {KNOBS[o.split("=")[0]]: int(o.split("=")[1]) for o in opts if "=" in o}
    q, k, v, o = tensors(B, H, N, D, bf16, zero)
    if vt:
        v = v.transpose(-2, -1).contiguous()
    for kk, vv in knobs.items():
        capi.tune(kk, vv)
    try:
        name = capi.attn_kernel_name(N, D, vt, bf16, bh=B * H)
        step = (lambda: capi.attn_fwd_bf16(q, k, v, o)) if bf16 else (lambda: capi.attn_fwd(q, k, v, o, v_transposed=vt))
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            step()
        e1.record()
        torch.cuda.synchronize()
        n = max(5, int(secs / (e0.elapsed_time(e1) / 5 * 1e-3)))
        e0.record()
        for _ in range(n):
            step()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
    finally:
        for kk in knobs:
            capi.tune(kk, capi.tune_items()[kk][1])      # back to the library default
    return name, host.mha_matmul_flops(B, H, N, D) / ms * 1e-9, ms


res = {s: [] for s in args}
names = {}
for r in range(rounds):
    for s in args:
        names[s], tf, ms = run(s)
        res[s].append(tf)
for s in args:
    v = sorted(res[s])
    print(f"RATE {s:34s} {names[s]:44s} median {v[len(v) // 2]:7.1f}  best {v[-1]:7.1f}  worst {v[0]:7.1f} TFLOP/s", flush=True)
