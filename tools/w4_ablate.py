"""hgemm_w4b (32x32x16 4-wave kernel) ablation timing; tools/sustain.py hgemm:abl=N does the same under power_watch.sh (diagnosis only; ablated variants compute WRONG results).
Needs the diagnostic kernel instantiations: `LC_DIAG=1 python -m leetcuda_amd.build --force` first."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from leetcuda_amd import capi, host
n = 8192
a = torch.randn(n, n, dtype=torch.half, device="cuda")
b = torch.randn(n, n, dtype=torch.half, device="cuda")
c = torch.empty(n, n, dtype=torch.half, device="cuda")
flops = 2.0 * n ** 3
for lname, lay in (("tn", capi.LAYOUT_TN), ("nn", capi.LAYOUT_NN)):
    b2 = host.as_col_major(b) if lay == capi.LAYOUT_TN else b
    for rnd in range(2):
        for abl in (0, 2, 4, 6, 8, 14):
            capi.tune("w4_abl", abl)
            ms = capi.hgemm_time(a, b2, c, lay, capi.HGEMM_MFMA256W4C, 2, 2048, warmup=2, iters=20)
            print(f"{lname} round {rnd} w4_abl {abl}: {ms:.4f} ms {flops / ms * 1e-9:8.1f} TF-equivalent", flush=True)
        capi.tune("w4_abl", 0)
        ms = capi.hgemm_time(a, b2, c, lay, capi.HGEMM_MFMA256P2, 2, 2048, warmup=2, iters=20)
        print(f"{lname} round {rnd} pingpong2: {ms:.4f} ms {flops / ms * 1e-9:8.1f} TFLOP/s", flush=True)
