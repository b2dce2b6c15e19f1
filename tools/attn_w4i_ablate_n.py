#!/usr/bin/env python3
"""Is the MFMA-only remainder of the attention stream a per-tile or a per-block cost?  The ablated copies of tools/attn_w4i_ablate.py
(liblc_diag.so, results WRONG by design) on zero-filled operands at growing sequence lengths: one 256-row block walks N / 64 tiles."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from leetcuda_amd import capi, host  # noqa: E402

capi.load()
diag = capi.load_diag()
NAMES = {0: "full", 7: "MFMA only", 23: "MFMA only, no barrier", 55: "MFMA only, no barrier, no guard"}
for B, H, N, D in ((4, 32, 4096, 128), (2, 16, 16384, 128), (1, 8, 65536, 128)):
    fl = host.mha_matmul_flops(B, H, N, D)
    q = torch.zeros(B, H, N, D, dtype=torch.half, device="cuda")
    k, v, o = torch.zeros_like(q), torch.zeros_like(q), torch.zeros_like(q)
    row = []
    for abl in (0, 7, 23, 55):
        if abl == 0:
            capi.tune("attn_nw", 514)
            step = lambda: capi.attn_fwd(q, k, v, o)  # noqa: E731
        else:
            step = lambda a=abl: diag.lc_diag_attn_w4i(a, q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), B, H, N, D, None)  # noqa: E731
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 20
        for _ in range(n):
            step()
        e1.record()
        torch.cuda.synchronize()
        capi.tune("attn_nw", 0)
        tf = fl / (e0.elapsed_time(e1) / n) * 1e-9
        row.append(f"{NAMES[abl]} {tf:7.1f} ({2500.0 / tf * 16:5.1f} cyc/MFMA at 2.4 GHz)")
    print(f"B{B} H{H} N{N}: " + " | ".join(row), flush=True)
