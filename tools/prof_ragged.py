#!/usr/bin/env python3
"""The ragged-shape paths of LC_HGEMM_AUTO under rocprofv3 (late round 6): per shape 30 launches of ours and of hipBLASLt, so that
`rocprofv3 --kernel-trace --stats` shows what each path launches and for how long (interior, border, pad copies, split-K + reduce).

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6al/prof -- python tools/prof_ragged.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from leetcuda_amd import capi, host  # noqa: E402

capi.load()
capi.vendor_init()
for (M, N, K) in ((8200, 8200, 8192), (1000, 3000, 4096), (8192, 8192, 8200), (100, 4096, 4096), (3000, 3000, 3008)):
    a = torch.randn(M, K, dtype=torch.half, device="cuda")
    b = torch.randn(K, N, dtype=torch.half, device="cuda")
    c = torch.empty(M, N, dtype=torch.half, device="cuda")
    bt = host.as_col_major(b)
    print(M, N, K, capi.hgemm_kernel_name(M, N, K, capi.LAYOUT_TN), flush=True)
    for _ in range(30):
        capi.hgemm(a, bt, c, layout=capi.LAYOUT_TN, variant=capi.HGEMM_AUTO, swizzle_stride=host.make_block_swizzle_stride(N, K))
    for _ in range(30):
        capi.hgemm_vendor(a, bt, c, capi.LAYOUT_TN)
    torch.cuda.synchronize()
capi.vendor_destroy()
