#!/usr/bin/env python3
"""Price the instruction classes of the generated attention stream (attn_fwd_w4i_kernel) on hardware: the full kernel against
copies with one class REMOVED from the phase statements (liblc_diag.so, lc_diag_attn_w4i; results WRONG by design).
  abl bits: 1 no LDS-DMA, 2 no LDS reads, 4 no softmax VALU, 8 no MFMA, 16 no per-tile wait + barrier, 32 no guard decision      (0 = the shipped kernel through the C-ABI)
usage: attn_w4i_ablate.py [--seconds S]      prints TFLOP/s-equivalent (matmul FLOPs / time) on randn and on zero-filled inputs"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from leetcuda_amd import capi, host  # noqa: E402

secs = float(sys.argv[sys.argv.index("--seconds") + 1]) if "--seconds" in sys.argv else 0.5
capi.load()
diag = capi.load_diag()


def rate(step, flops):
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
    return flops / (e0.elapsed_time(e1) / n) * 1e-9


NAMES = {0: "full", 1: "no DMA", 2: "no LDS reads", 3: "no DMA, no reads", 4: "no softmax VALU", 7: "MFMA only", 8: "no MFMA",
         23: "MFMA only, no barrier", 55: "MFMA only, no barrier, no guard"}
for shape in ((4, 32, 4096, 128), (1, 48, 8192, 64)):
    B, H, N, D = shape
    fl = host.mha_matmul_flops(B, H, N, D)
    for fillname in ("randn", "zero"):
        torch.manual_seed(0)
        mk = (lambda: torch.zeros(B, H, N, D, dtype=torch.half, device="cuda")) if fillname == "zero" else \
             (lambda: torch.randn(B, H, N, D, dtype=torch.half, device="cuda"))
        q, k, v = mk(), mk(), mk()
        o = torch.zeros_like(q)
        row = []
        for abl in (0, 1, 2, 3, 4, 7, 8, 23, 55):
            if abl == 0:
                capi.tune("attn_nw", 514)
                try:
                    r = rate(lambda: capi.attn_fwd(q, k, v, o), fl)
                finally:
                    capi.tune("attn_nw", 0)
            else:
                def step(a=abl):
                    rc = diag.lc_diag_attn_w4i(a, q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), B, H, N, D,
                                               torch.cuda.current_stream().cuda_stream)
                    assert rc == 0, rc
                r = rate(step, fl)
            row.append(f"{NAMES[abl]} {r:7.1f}")
        print(f"ABL {shape} {fillname:5s}: " + " | ".join(row), flush=True)
