#!/usr/bin/env python3
"""debug: hgemm_mid_kernel combos vs torch fp32 matmul; prints per-16x16-block error map for failing cases"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from leetcuda_amd import capi, host
capi.load()
for lname, lay in (("tn", capi.LAYOUT_TN), ("nn", capi.LAYOUT_NN)):
    for tmw in (1, 2):
        for tnw in ((2, 3) if lname == "tn" else (2,)):
            for ns in (2, 3):
                capi.tune("hgemm_mid", 10 * tmw + tnw); capi.tune("hgemm_mid_ns", ns)
                tm, tn = 64 * tmw, 64 * tnw
                for (M, N, K) in [(tm, tn, 64), (tm, tn, 128), (tm, tn, 192), (tm, tn, 256), (2 * tm, 2 * tn, 512), (tm, tn, 96)]:
                    torch.manual_seed(1)
                    a = torch.randn(M, K, dtype=torch.half, device="cuda")
                    b = torch.randn(K, N, dtype=torch.half, device="cuda")
                    bb = host.as_col_major(b) if lay == capi.LAYOUT_TN else b
                    c = torch.full((M, N), float("nan"), dtype=torch.half, device="cuda")
                    capi.hgemm(a, bb, c, layout=lay, variant=capi.HGEMM_MID, swizzle_stride=1)
                    torch.cuda.synchronize()
                    ref = a.float() @ b.float()
                    err = (c.float() - ref).abs()
                    bad = err > 0.05 + 0.01 * ref.abs()
                    print(f"{lname} tmw {tmw} tnw {tnw} ns {ns} {M}x{N}x{K}: max err {err.max().item():.3f} bad {bad.float().mean().item():.3f} nan {torch.isnan(c).float().mean().item():.3f}", flush=True)
                    if bad.any() and M <= 128:
                        blk = bad.reshape(M // 16, 16, N // 16, 16).any(dim=3).any(dim=1).int().cpu().numpy()
                        print(blk)
capi.tune("hgemm_mid", 0); capi.tune("hgemm_mid_ns", 0)
