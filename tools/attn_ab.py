#!/usr/bin/env python3
"""Interleaved A/B of the D = 128 attention kernels at config 3 and config 4's per-rank shard (lc_tune_set "attn_nw")."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from leetcuda_amd import capi, host  # noqa: E402

capi.load()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for (B, H, N, D) in ((4, 32, 4096, 128), (4, 32, 8192, 128)):
    q, k, v, o, _ = host.get_qkvo(B, H, N, D, seed=0)
    fl = host.mha_matmul_flops(B, H, N, D)
    ref = None
    for nw in (0, 513, 515, 517, 8):
        capi.tune("attn_nw", nw)
        capi.attn_fwd(q, k, v, o)
        torch.cuda.synchronize()
        if ref is None:
            ref = o.clone()
        else:
            print(f"B{B} S{N} nw={nw}: max |o - o_default| = {(o.float() - ref.float()).abs().max().item():.3e}", flush=True)
    for r in range(rounds):
        for nw in (0, 513, 515, 517, 8):
            capi.tune("attn_nw", nw)
            ms = capi.attn_time(q, k, v, o, False, capi.ATTN_SPLIT_Q, 2, warmup=2, iters=10)
            print(f"B{B} S{N} round {r} nw={nw:3d} {capi.attn_kernel_name(N, D):34s}: {ms:.4f} ms {fl / ms * 1e-9:8.1f} TFLOP/s", flush=True)
    capi.tune("attn_nw", 0)
    del q, k, v, o
