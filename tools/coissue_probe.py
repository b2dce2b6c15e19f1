#!/usr/bin/env python3
"""What does an instruction cost next to an MFMA stream on one SIMD?  (lc_probe_coissue, MI355X)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from leetcuda_amd import capi  # noqa: E402

lib = capi.load_diag()
out = torch.zeros(16, dtype=torch.int64, device="cuda")
names = {1: "v_fma_f32", 2: "v_exp_f32", 3: "v_pk_fma_f32", 4: "v_cvt_pk_f16_f32", 5: "ds_read_b128",
         6: "v_accvgpr_read", 7: "v_exp+dep v_add", 8: "ds_read_b64_tr_b16"}


def run(f, k, mode):
    out.zero_()
    for _ in range(2):
        rc = lib.lc_probe_coissue(f, k, mode, out.data_ptr(), None)
        assert rc == 0, rc
        torch.cuda.synchronize()
    return out.cpu().numpy()


base = run(0, 1, 0)
print(f"MFMA only, 1 wave/SIMD: {base[0] / 1024:.1f} cycles per MFMA")
base3 = run(0, 1, 3)
print(f"MFMA only, AGPR accumulator + AGPR B operand: {base3[0] / 1024:.1f} cycles per MFMA")
for f in (1, 2, 3, 4, 5, 6, 7, 8):
    for k in (1, 2, 4, 8):
        own = run(f, k, 0)[0] / 1024
        own3 = run(f, k, 3)[0] / 1024
        alone = run(f, k, 2)[0] / 1024
        print(f"{names[f]:18s} k={k}: beside VGPR-MFMA {own:6.1f} cyc/MFMA (+{(own - base[0] / 1024) / k:5.1f} per filler) | "
              f"beside AGPR-MFMA {own3:6.1f} (+{(own3 - base3[0] / 1024) / k:5.1f}) | fillers alone {alone / k:5.1f} each")
