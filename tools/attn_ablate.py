#!/usr/bin/env python3
"""Ablation timing of the lock-step attention kernel at config 3 (results are wrong by design when a bit
is set): which phase carries the time? bits: 1 no exp, 2 no PV, 4 no QK, 8 no staging, 16 no barrier.
Needs the diagnostic kernel instantiations: `LC_DIAG=1 python -m leetcuda_amd.build --force` first."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from leetcuda_amd import capi, host  # noqa: E402

capi.load()
q, k, v, o, _ = host.get_qkvo(4, 32, 4096, 128, seed=0)
fl = host.mha_matmul_flops(4, 32, 4096, 128)
capi.tune("attn_nw", 8)
for rnd in range(2):
    for abl in (0, 1, 2, 3, 4, 6, 7, 8, 16, 24, 30, 31):
        capi.tune("attn_ablate", abl)
        ms = capi.attn_time(q, k, v, o, False, capi.ATTN_SPLIT_Q, 2, warmup=2, iters=10)
        print(f"round {rnd} ablate {abl:2d}: {ms:.4f} ms ({fl / ms * 1e-9:7.1f} TF-equivalent)", flush=True)
capi.tune("attn_ablate", 0)
capi.tune("attn_nw", 0)
