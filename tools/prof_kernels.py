#!/usr/bin/env python3
"""Launch the hot kernels a few times with BASELINE-sized inputs — the target command for
`rocprofv3 --kernel-trace --stats` and the separate `--pmc` passes (profiles/README.md)."""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from leetcuda_amd import capi, host  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--what", default="all", choices=["all", "hgemm", "attn"])
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--stagger", type=lambda v: int(v, 0), default=None, help="lc_tune_set hgemm_stagger for the GEMM launches")
ap.add_argument("--only-auto", action="store_true", help="GEMM: only the default kernel + hipBLASLt")
a = ap.parse_args()
capi.load()
torch.manual_seed(0)
if a.what in ("all", "hgemm"):
    n = 8192
    A = torch.randn(n, n, dtype=torch.half, device="cuda")
    B = torch.randn(n, n, dtype=torch.half, device="cuda")
    C = torch.zeros(n, n, dtype=torch.half, device="cuda")
    Bt = host.as_col_major(B)
    if a.stagger is not None:
        capi.tune("hgemm_stagger", a.stagger or 1 << 27)     # (0 on the command line = off)
    for var in ((capi.HGEMM_MFMA256W4Y,) if a.only_auto else
                (capi.HGEMM_MFMA256W4Y, capi.HGEMM_MFMA256W4X, capi.HGEMM_MFMA256W4C, capi.HGEMM_MFMA256P2, capi.HGEMM_MFMA256)):
        for lay, bb in ((capi.LAYOUT_TN, Bt), (capi.LAYOUT_NN, B)):
            for _ in range(a.iters):
                capi.hgemm(A, bb, C, layout=lay, variant=var, swizzle_stride=2048)
    capi.vendor_init()   # hipBLASLt on the same inputs: its kernels show up under their own (Cijk_...) names
    for lay, bb in ((capi.LAYOUT_TN, Bt), (capi.LAYOUT_NN, B)):
        for _ in range(a.iters):
            capi.hgemm_vendor(A, bb, C, layout=lay)
    torch.cuda.synchronize()
    # round 6: the mid-size kernel at 2048^3 (hgemm_mid_kernel<.,2,2,3>: 256 workgroups of 128 x 128, one round; tools/prof_workloads.py
    # "hgemm_2048") next to hipBLASLt's MT128x128x64 kernel on the same operands
    m = 2048
    A2, B2, C2 = A[:m, :m].contiguous(), B[:m, :m].contiguous(), torch.zeros(m, m, dtype=torch.half, device="cuda")
    Bt2 = host.as_col_major(B2)
    for lay, bb in ((capi.LAYOUT_TN, Bt2), (capi.LAYOUT_NN, B2)):
        for _ in range(4 * a.iters):
            capi.hgemm(A2, bb, C2, layout=lay, variant=capi.HGEMM_AUTO, swizzle_stride=1024)
        for _ in range(4 * a.iters):
            capi.hgemm_vendor(A2, bb, C2, layout=lay)
    torch.cuda.synchronize()
    del A2, B2, C2, Bt2
    # late round 6: a ragged shape (LC_HGEMM_RAGGED; tools/prof_workloads.py "hgemm_ragged"): the whole of 2000 x 2000 x 2048 on hgemm_mid_edge_kernel<.,2,2,3>
    # (256 clamped tiles, one round).  No flagship-kernel shape and no vendor launch here: their per-launch averages belong to the 8192^3 / 2048^3 workloads.
    A3 = torch.randn(2000, 2048, dtype=torch.half, device="cuda")
    B3 = torch.randn(2048, 2000, dtype=torch.half, device="cuda")
    C3 = torch.zeros(2000, 2000, dtype=torch.half, device="cuda")
    Bt3 = host.as_col_major(B3)
    for lay, bb in ((capi.LAYOUT_TN, Bt3), (capi.LAYOUT_NN, B3)):
        for _ in range(4 * a.iters):
            capi.hgemm(A3, bb, C3, layout=lay, variant=capi.HGEMM_AUTO, swizzle_stride=1024)
    torch.cuda.synchronize()
    del A3, B3, C3, Bt3
if a.what in ("all", "attn"):
    q, k, v, o, tv = host.get_qkvo(4, 32, 4096, 128, seed=0)
    for nw in (0, 517, 514, 8):     # default (persistent merged-phase kernel, static walk), the dynamic-queue walk, the generated one-statement-per-phase twin, lock-step
        capi.tune("attn_nw", nw)
        capi.tune("attn_w4i_sched", 1 if nw == 514 else 0)
        for _ in range(a.iters):
            capi.attn_fwd(q, k, v, o)
    capi.tune("attn_nw", 0)
    capi.tune("attn_w4i_sched", 1)
    torch.cuda.synchronize()
    tv2 = v.transpose(-2, -1).contiguous()     # the V-transposed entries at config 3
    for _ in range(a.iters):
        capi.attn_fwd(q, k, tv2, o, v_transposed=True)
    torch.cuda.synchronize()
    del q, k, v, o, tv, tv2
    # config 4 (B32 H32 S8192 D128: 2 GiB per tensor): attn_fwd_w4u_kernel<128,false,0> — one block per workgroup (prof_workloads: "attn_cfg4")
    q, k, v, o, tv = host.get_qkvo(32, 32, 8192, 128, seed=0)
    for _ in range(2):
        capi.attn_fwd(q, k, v, o)
    torch.cuda.synchronize()
    del q, k, v, o, tv
    # split-KV (round 5): (1,4,4096,128) = 64 query blocks on 256 CUs -> attn_fwd_w4u_kernel<128,false,3> with 4 KV ranges per block +
    # attn_split_combine_kernel<128> (tools/prof_workloads.py: "attn_split_1x4x4096")
    q, k, v, o, tv = host.get_qkvo(1, 4, 4096, 128, seed=0)
    for _ in range(4 * a.iters):
        capi.attn_fwd(q, k, v, o)
    torch.cuda.synchronize()
    del q, k, v, o, tv
    # the reference's published shape (1,48,8192,64) (tools/prof_workloads.py: "attn_d64")
    q, k, v, o, tv = host.get_qkvo(1, 48, 8192, 64, seed=0)
    for _ in range(a.iters):
        capi.attn_fwd(q, k, v, o)
    torch.cuda.synchronize()
    del q, k, v, o, tv
    # config 5a: FFPA shape, fp16 through the tiling-QKV entry and bf16 (full-width kernel attn_bigd2.hip)
    q = torch.randn(1, 48, 8192, 512, device="cuda").half()
    k = torch.randn(1, 48, 8192, 512, device="cuda").half()
    v = torch.randn(1, 48, 8192, 512, device="cuda").half()
    o = torch.zeros_like(q)
    for _ in range(max(1, a.iters // 2)):
        capi.attn_call("flash_attn_mma_stages_split_q_tiling_qkv", q, k, v, o, 2)
    qb, kb, vb, ob = q.bfloat16(), k.bfloat16(), v.bfloat16(), o.bfloat16()
    for _ in range(max(1, a.iters // 2)):
        capi.attn_fwd_bf16(qb, kb, vb, ob)
    torch.cuda.synchronize()
    del q, k, v, o, qb, kb, vb, ob
    # D = 256 (1,48,8192,256): attn_bigd7.hip, fp16 and bf16 (tools/prof_workloads.py: "attn_d256_fp16" / "attn_d256_bf16")
    q, k, v, o, tv = host.get_qkvo(1, 48, 8192, 256, seed=0)
    for _ in range(max(1, a.iters // 2)):
        capi.attn_call("flash_attn_mma_stages_split_q_tiling_qkv", q, k, v, o, 2)
    qb, kb, vb, ob = q.bfloat16(), k.bfloat16(), v.bfloat16(), o.bfloat16()
    for _ in range(max(1, a.iters // 2)):
        capi.attn_fwd_bf16(qb, kb, vb, ob)
    torch.cuda.synchronize()
    del q, k, v, o, tv, qb, kb, vb, ob
    # D = 1024 (1,48,8192,1024): the pair kernel attn_bigd4.hip (tools/prof_workloads.py: "attn_d1024")
    q, k, v, o, tv = host.get_qkvo(1, 48, 8192, 1024, seed=0)
    for _ in range(max(1, a.iters // 2)):
        capi.attn_call("flash_attn_mma_stages_split_q_tiling_qkv", q, k, v, o, 2)
    torch.cuda.synchronize()
    del q, k, v, o, tv
    n = 16384   # config 5b: fp8 e4m3 GEMM (tools/prof_workloads.py: "fp8_16384")
    a8 = torch.randn(n, n, device="cuda").to(torch.float8_e4m3fn)
    b8 = torch.randn(n, n, device="cuda").to(torch.float8_e4m3fn)
    c8 = torch.zeros(n, n, dtype=torch.half, device="cuda")
    for _ in range(a.iters):
        capi.gemm_fp8(a8, b8, c8, alpha=1 / 16, swizzle_stride=2048)
    torch.cuda.synchronize()
print("done")
