#!/usr/bin/env python3
"""WAR window of an in-flight MFMA's A operand (lc_probe_mfma_war)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from leetcuda_amd import capi  # noqa: E402
lib = capi.load_diag()
torch.manual_seed(0)
a = torch.randn(32, 16, device="cuda").half(); b = torch.randn(32, 16, device="cuda").half()
want = (b.float() @ a.float().t())   # d[row from first operand = a? see probe_mfma32: rows follow operand 1]
d = torch.zeros(32, 32, device="cuda")
for kind, queued in ((0, 0), (0, 1), (0, 4), (1, 0), (1, 1), (2, 0), (2, 1), (2, 2), (2, 4)):
    if True:
        res = []
        for delay in (0, 1, 2, 3, 4, 6, 8, 11, 15):
            d.zero_()
            assert lib.lc_probe_mfma_war(delay, kind, queued, a.data_ptr(), b.data_ptr(), d.data_ptr(), None) == 0
            torch.cuda.synchronize()
            w1 = a.float() @ b.float().t()
            err = min((d - w1).abs().max().item(), (d - w1.t()).abs().max().item())
            res.append(f"{delay}:{'ok' if err < 1e-2 else f'BAD({err:.1f})'}")
        print(f"overwrite by {('v_mov', 'v_exp', 'ds_read_b128')[kind]}, {queued} MFMA(s) ahead in the pipe: " + " ".join(res))
