#!/usr/bin/env python3
"""Cycle stamps of the two-phase ping-pong HGEMM (diagnosis; lc_tune_set hgemm_stamps=1 clobbers A).
Needs the diagnostic kernel instantiations: `LC_DIAG=1 python -m leetcuda_amd.build --force` first."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from leetcuda_amd import capi, host  # noqa: E402
capi.load()
n = 8192
a = torch.randn(n, n, dtype=torch.half, device="cuda")
b = torch.randn(n, n, dtype=torch.half, device="cuda")
c = torch.zeros(n, n, dtype=torch.half, device="cuda")
bt = host.as_col_major(b)
for lay, bb, nm in ((1, bt, "tn"), (0, b, "nn")):
    for _ in range(5):
        capi.hgemm(a, bb, c, layout=lay, variant=4, swizzle_stride=2048)
    torch.cuda.synchronize()
    a2 = a.clone()
    capi.tune("hgemm_stamps", 1)
    capi.hgemm(a2, bb, c, layout=lay, variant=4, swizzle_stride=2048)
    torch.cuda.synchronize()
    capi.tune("hgemm_stamps", 0)
    st = a2.view(-1)[:2 * 4 * 8 * 4].view(torch.int64).cpu().numpy().reshape(2, 4, 8).astype("int64")
    t00 = st[0, 0, 0]
    for w in range(2):
        for t in range(4):
            r = st[w, t]
            print(f"{nm} wave{w*4} kt{32+t}: top={int(r[0]-t00):6d} readsA={int(r[5]-r[0]):4d} dmawaitA={int(r[1]-r[5]):4d} barA={int(r[2]-r[1]):4d} "
                  f"mfmaA={int(r[3]-r[2]):4d} barA2={int(r[4]-r[3]):4d} loadB+barB={int(r[6]-r[4]):4d} "
                  f"mfmaB={int(r[7]-r[6]):4d}" + (f" barB2+loop={int(st[w,t+1,0]-r[7]):4d}" if t < 3 else ""))
