#!/usr/bin/env python3
"""Does an attention kernel's result depend on state it did not initialise?  Before every launch a polluter kernel
(liblc_diag.so: lc_diag_pollute) leaves a bit pattern in every VGPR / AGPR / LDS byte of the chip; the outputs of runs that
follow DIFFERENT patterns must be bit-identical.  usage: attn_determinism.py [nw ...]   (attn_nw values, default 512 256 8)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from leetcuda_amd import capi  # noqa: E402

capi.load()
diag = capi.load_diag()
B, H, N, D = 4, 32, 8192, 128
torch.manual_seed(4)
q = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
k = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
v = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
PATTERNS = {"zero": 0x00000000, "nan": 0x7fc07e00, "ones": 0xffffffff, "big": 0x7bff7bff, "one": 0x3c003c00}
WHAT = {"vgpr": 1, "agpr": 2, "lds": 4, "all": 7}
nws = [int(x) for x in sys.argv[1:]] or [512, 256, 8]
for nw in nws:
    capi.tune("attn_nw", nw)
    ref = None
    for wname, what in WHAT.items():
        for pname, pat in PATTERNS.items():
            assert diag.lc_diag_pollute(pat, what, None) == 0
            o = torch.full_like(q, float("nan"))
            capi.attn_fwd(q, k, v, o)
            torch.cuda.synchronize()
            if ref is None:
                ref = o
                continue
            if not torch.equal(ref, o):
                bad = (ref != o).nonzero().cpu()
                rows = sorted(set((int(x[0]), int(x[1]), int(x[2])) for x in bad))
                nanc = int(torch.isnan(o).sum())
                print(f"nw={nw} after pollute({wname}, {pname}): {len(bad)} elements / {len(rows)} rows differ, max "
                      f"{(ref.float() - o.float()).abs().nan_to_num(1e9).max().item():.3e}, NaNs {nanc}; first rows {rows[:4]}", flush=True)
    print(f"nw={nw} ({capi.attn_kernel_name(N, D)}): done", flush=True)
capi.tune("attn_nw", 0)
