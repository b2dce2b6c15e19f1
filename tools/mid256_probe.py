#!/usr/bin/env python3
"""hgemm_mid_kernel's loop on a 256 x 256 tile (liblc_diag.so lc_probe_mid256) against hgemm_w4y_kernel and hipBLASLt: is the rotated
hand-ordered loop as good as the generated one?   tools/mid256_probe.py [sizes] [seconds]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from leetcuda_amd import capi, host  # noqa: E402

capi.load()
diag = capi.load_diag()
capi.vendor_init()
sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [4096, 8192]
SECONDS = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6


def burst(step, n):
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n):
        step()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) * 1e-3


for n in sizes:
    a = torch.randn(n, n, dtype=torch.half, device="cuda")
    b = torch.randn(n, n, dtype=torch.half, device="cuda")
    c = torch.empty(n, n, dtype=torch.half, device="cuda")
    fl = 2.0 * n ** 3
    st = host.make_block_swizzle_stride(n, n)
    for lname, lay in (("tn", capi.LAYOUT_TN), ("nn", capi.LAYOUT_NN)):
        b2 = host.as_col_major(b) if lay == capi.LAYOUT_TN else b
        stream = torch.cuda.current_stream().cuda_stream
        pw = max(1, st // 256)

        def mid():
            rc = diag.lc_probe_mid256(a.data_ptr(), b2.data_ptr(), c.data_ptr(), n, n, n, int(lay == capi.LAYOUT_NN), pw, stream)
            assert rc == 0, rc
        mid(); torch.cuda.synchronize()
        ref = torch.empty_like(c)
        capi.hgemm(a, b2, ref, layout=lay, variant=capi.HGEMM_MFMA256W4Y, swizzle_stride=st)
        torch.cuda.synchronize()
        err = (c.float() - ref.float()).abs().max().item()
        cands = {"mid256": mid, "w4y": lambda: capi.hgemm(a, b2, c, layout=lay, variant=capi.HGEMM_MFMA256W4Y, swizzle_stride=st),
                 "w4y nostagger": None, "hipBLASLt": lambda: capi.hgemm_vendor(a, b2, c, lay)}

        def nost():
            capi.tune("hgemm_stagger", 1 << 27); capi.tune("hgemm_persist", 0)
            capi.hgemm(a, b2, c, layout=lay, variant=capi.HGEMM_MFMA256W4Y, swizzle_stride=st)
            capi.tune("hgemm_stagger", 0); capi.tune("hgemm_persist", 1)
        cands["w4y nostagger"] = nost
        for f in cands.values():
            burst(f, 2)
        per = burst(cands["w4y"], 5) / 5
        cnt = max(5, int(SECONDS / 3 / per))
        t = {k: 0.0 for k in cands}
        keys = list(cands)
        for r in range(3):
            for k in keys[r:] + keys[:r]:
                t[k] += burst(cands[k], cnt)
        print(f"n={n:5d} {lname} (max |mid256 - w4y| = {err:.3g}): " + " | ".join(f"{k} {fl * cnt * 3 / v * 1e-12:6.1f}" for k, v in t.items()), flush=True)
capi.vendor_destroy()
