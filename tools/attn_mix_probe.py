#!/usr/bin/env python3
"""One wave per SIMD or two?  Cycles per v_mfma_f32_16x16x32_f16 and SIMD next to the softmax share of an MFMA slot
(lc_probe_attn_mix in liblc_diag.so).  16 = the matrix core flat out."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from leetcuda_amd import capi  # noqa: E402

lib = capi.load_diag()
out = torch.zeros(16, dtype=torch.int64, device="cuda")
names = {0: "MFMAs only", 1: "D = 128 share (exp + add + fma + cvt/2 + ds_read/2 per two slots)", 2: "D = 64 share (the same per slot)",
         3: "D = 64 share without the LDS read", 4: "D = 128 share without the LDS read",
         5: "D = 64 share, exp2 as a packed-fp16 polynomial (10 VALU per 2 scores)", 6: "D = 128 share, exp2 as a packed-fp16 polynomial",
         7: "D = 64 share, Q.K^T on v_mfma_f32_32x32x16 (re-layout NOT charged)", 8: "D = 128 share, Q.K^T on v_mfma_f32_32x32x16 (re-layout NOT charged)"}
for mix in (0, 1, 2, 3, 4, 5, 6, 7, 8):
    row = []
    for waves in (4, 8):
        for _ in range(3):
            out.zero_()
            rc = lib.lc_probe_attn_mix(waves, mix, out.data_ptr(), None)
            assert rc == 0, rc
            torch.cuda.synchronize()
        t = out.cpu().numpy()[:waves]
        # (mixes 7 / 8: one 32x32x16 MFMA counts as the TWO 16x16x32 slots whose FLOPs it does: cycles per slot-equivalent)
        row.append(f"{waves} waves: {t.max() / 2048 / (waves // 4):5.1f} cycles per MFMA and SIMD (per wave {t.max() / 2048:5.1f})")
    print(f"{names[mix]:72s} | " + " | ".join(row), flush=True)

forms = {0: "A, B, C/D in VGPRs", 1: "A, B in AGPRs, C/D in VGPRs (Q.K^T form)", 2: "A, B in VGPRs, C/D in AGPRs (P.V form)",
         3: "forms 1 and 2 alternating (merged phase)", 4: "A, B, C/D in AGPRs"}
for form in range(5):
    for _ in range(3):
        out.zero_()
        rc = lib.lc_probe_mfma_form(form, out.data_ptr(), None)
        assert rc == 0, rc
        torch.cuda.synchronize()
    t = out.cpu().numpy()[:4]
    print(f"MFMA operand files: {forms[form]:48s} {t.max() / 4096:5.1f} cycles per MFMA", flush=True)
