#!/usr/bin/env python3
"""Step cycle stamps of the AUTO HGEMM kernel (hgemm_w4b_kernel<.., BUF=true>; lc_tune_set hgemm_stamps=1 clobbers A).
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
        capi.hgemm(a, bb, c, layout=lay, variant=capi.HGEMM_MFMA256W4C, swizzle_stride=2048)
    torch.cuda.synchronize()
    a2 = a.clone()
    capi.tune("hgemm_stamps", 1)
    capi.hgemm(a2, bb, c, layout=lay, variant=capi.HGEMM_MFMA256W4C, swizzle_stride=2048)
    torch.cuda.synchronize()
    capi.tune("hgemm_stamps", 0)
    st = a2.view(-1)[:4 * 8 * 4].view(torch.int64).cpu().numpy().reshape(4, 8).astype("int64")
    for t in range(4):
        r = st[t]
        nxt = int(st[t + 1, 0] - r[6]) if t < 3 else -1
        print(f"{nm} kt{32+t}: top={int(r[0]-st[0,0]):6d} step0={int(r[1]-r[0]):4d} step1(+4 DMA)={int(r[2]-r[1]):4d} "
              f"step2(+4 DMA)={int(r[3]-r[2]):4d} wait={int(r[4]-r[3]):4d} barrier={int(r[5]-r[4]):4d} "
              f"step3(+8 DMA)={int(r[6]-r[5]):4d} loop={nxt}")
