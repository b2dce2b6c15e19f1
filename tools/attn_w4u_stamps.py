#!/usr/bin/env python3
"""Where does a block of attn_fwd_w4u_kernel<128, false, *> spend its time?  Cycle stamps of wave 0 of workgroup 0 (liblc_diag.so,
lc_diag_attn_w4u_stamps) for a few shapes / split factors: DESIGN.md section 9 item 1 (12 us fixed per block against 0.96 us per KV tile).
usage: attn_w4u_stamps.py [B,H,N[:S]] ..."""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from leetcuda_amd import capi  # noqa: E402

lib = capi.load_diag()
NAMES = ["entry", "tiles 0,1 + Q requested", "Q parked in AGPRs, O zeroed", "tiles landed + barrier", "first S^T + row max", "tile 0 (2 phases)",
         "tiles 1 .. T-2", "last tile + tail P.V", "drain + epilogue barrier", "O staged in LDS", "O stores issued", "O stores acknowledged"]
specs = sys.argv[1:] or ["1,8,1024", "1,8,1024:4", "1,8,1024:8", "1,4,4096", "1,4,4096:4", "4,32,4096"]
for spec in specs:
    shape, *rest = spec.split(":")
    B, H, N = (int(x) for x in shape.split(","))
    S = int(rest[0]) if rest else 1
    D = 128
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, N, D, dtype=torch.half, device="cuda") for _ in range(3))
    o = torch.zeros(S, B, H, N, D, dtype=torch.half, device="cuda")
    lse = torch.zeros(S, B, H, N, dtype=torch.float32, device="cuda")
    out = (C.c_ulonglong * 16)()
    best = None
    for _ in range(5):      # the last of a few runs: clocks up, code in L2
        rc = lib.lc_diag_attn_w4u_stamps(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), B, H, N, S, C.addressof(out),
                                         torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
        best = list(out)
    t = best
    total = t[11] - t[0]
    us = (t[15] - t[14]) / 100.0
    ghz = total / (us * 1e3) if us > 0 else float("nan")
    print(f"== (B,H,N,D) = ({B},{H},{N},{D}), {S} KV range(s) per block, {N // 64 // S} tiles per workgroup, {(N // 256) * B * H * S} workgroups: "
          f"{total} cycles = {us:.2f} us in-kernel at {ghz:.2f} GHz", flush=True)
    for i in range(1, 12):
        d = t[i] - t[i - 1]
        print(f"   {NAMES[i]:32s} {d:8d} cycles  {d / (ghz * 1e3):6.2f} us" if ghz == ghz else f"   {NAMES[i]:32s} {d:8d} cycles", flush=True)
