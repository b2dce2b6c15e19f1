#!/usr/bin/env python3
"""Time the attention schedule variants (lc_tune_set "attn_nw") at config 3: 8 = lock-step, 16 = ping-pong,
32 = software-pipelined, 64 = LDS-DMA staged (when built)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from leetcuda_amd import capi, host  # noqa: E402

capi.load()
B, H, N, D = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (4, 32, 4096, 128)))
variants = [int(x) for x in sys.argv[5].split(",")] if len(sys.argv) > 5 else [8, 16, 32]
q, k, v, o, _ = host.get_qkvo(B, H, N, D, seed=0)
fl = host.mha_matmul_flops(B, H, N, D)
for rnd in range(3):
    for nw in variants:
        try:
            capi.tune("attn_nw", nw)
        except Exception:
            continue
        ms = capi.attn_time(q, k, v, o, False, capi.ATTN_SPLIT_Q, 2, warmup=3, iters=20)
        print(f"round {rnd} attn_nw {nw:2d}: {ms:.4f} ms {fl / ms * 1e-9:7.1f} TFLOP/s", flush=True)
capi.tune("attn_nw", 0)
