#!/usr/bin/env python3
"""hgemm_bench.py — the reference's HGEMM bench driver (kernels/hgemm/hgemm.py) on top of the drop-in module.

Same call convention (`hgemm.<entry>(a, b, c[, stages, swizzle, swizzle_stride])`), same timing method
(warm-up, `iters` launches between torch.cuda.synchronize(), wall clock; hgemm.py:247-272), same TFLOPS formula
(2MNK/t, :282) and the same output line (`tag: [first, last], time, swizzle<block>, TFLOPS(+x%)`, :288-304), for
the entry groups behind its CLI flags (--mma / --mma-all / --mma-tn / --cute-tn / --wmma / --wmma-all / --cuda /
--cuda-all, hgemm.py:469-1088).  Module lookup order is the reference's: `import toy_hgemm` first.

    PYTHONPATH=leetcuda_amd python tools/hgemm_bench.py --mma --mma-tn --cute-tn --MNK 8192
"""
import argparse
import sys
import time
from functools import partial
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "leetcuda_amd"))
import torch  # noqa: E402

from leetcuda_amd.host import as_col_major, make_block_swizzle_stride  # noqa: E402

ap = argparse.ArgumentParser()
for f in ("M", "N", "K", "MNK"):
    ap.add_argument(f"--{f}", type=int, default=None)
ap.add_argument("--MMNK", type=int, default=8192)
ap.add_argument("--SEP", type=int, default=4096)
ap.add_argument("--warmup", "--w", type=int, default=2)
ap.add_argument("--iters", "--i", type=int, default=10)
ap.add_argument("--swizzle-factor", "--sf", type=float, default=None)
for f in ("mma", "mma-all", "mma-tn", "cute-tn", "wmma", "wmma-all", "cuda", "cuda-all", "torch", "no-cublas",
          "show-all-info"):
    ap.add_argument(f"--{f}", action="store_true")
args = ap.parse_args()

import toy_hgemm as hgemm  # noqa: E402  (the reference's preferred import, tools/utils.py:131)

MAX_TFLOPS = -1.0


@torch.no_grad()
def run_benchmark(perf_func, a, b, tag, out=None, stages=-1, swizzle=False, swizzle_stride=1):
    global MAX_TFLOPS
    M, K = a.shape
    N = b.size(1)
    if swizzle:
        swizzle_stride = make_block_swizzle_stride(N, K, args.swizzle_factor)
        swizzle = swizzle if swizzle_stride >= 256 else False
    else:
        swizzle_stride = 1
    if out is not None:
        out.fill_(0)
    if "cublas" in tag:
        hgemm.init_cublas_handle()
    call = (lambda: perf_func(a, b, out, stages, swizzle, swizzle_stride)) if (out is not None and stages > 1) \
        else (lambda: perf_func(a, b, out)) if out is not None else (lambda: perf_func(a, b))
    for _ in range(args.warmup):
        call()
    torch.cuda.synchronize()
    start = time.time()
    for _ in range(args.iters):
        r = call()
    torch.cuda.synchronize()
    mean = (time.time() - start) / args.iters
    res = out if out is not None else r
    flat = res.flatten()
    vals = [f"{round(flat[0].item(), 8):<12}"[:10], f"{round(flat[-1].item(), 8):<12}"[:10]]
    tflops = (2 * M * N * K) * 1e-12 / mean
    ss = "NOOP" if swizzle_stride == 1 else swizzle_stride
    line = f"{tag:>53}: {vals}, time:{str(f'{mean * 1000:<12}')[:8]}ms, swizzle<block>: {ss:<4}, TFLOPS: {tflops:<6.2f}"
    if tflops > MAX_TFLOPS:
        imp = round((tflops - MAX_TFLOPS) / MAX_TFLOPS * 100, 2) if MAX_TFLOPS > 0 else 0
        MAX_TFLOPS = tflops
        print(f"{line}(+{imp:.2f}%)")
    elif args.show_all_info or "cublas" in tag:
        print(line)
    if "cublas" in tag:
        hgemm.destroy_cublas_handle()
    return res


def sizes():
    if args.M and args.N and args.K:
        return [(args.M, args.N, args.K)]
    if args.MNK:
        return [(args.MNK,) * 3]
    return [(s, s, s) for s in range(args.SEP, args.MMNK + args.SEP, args.SEP)]


P = "hgemm_mma_m16n8k16_"
for M, N, K in sizes():
    MAX_TFLOPS = -1.0
    print("-" * 150)
    print(f"M={M}, N={N}, K={K}, Warmup={args.warmup}, Iters={args.iters}".center(150))
    print("-" * 150)
    a = torch.randn((M, K), dtype=torch.half, device="cuda")
    b = torch.randn((K, N), dtype=torch.half, device="cuda")
    c = torch.randn((M, N), dtype=torch.half, device="cuda")
    bc = as_col_major(b)
    if args.cuda or args.cuda_all:
        run_benchmark(hgemm.hgemm_naive_f16, a, b, "(naive)", c)
        run_benchmark(hgemm.hgemm_t_8x8_sliced_k_f16x8_pack_bcf, a, b, "(f16x8pack+t8x8+bcf)", c)
        run_benchmark(hgemm.hgemm_t_8x8_sliced_k_f16x8_pack_bcf_dbuf, a, b, "(f16x8pack+t8x8+dbuf)", c)
        run_benchmark(hgemm.hgemm_t_8x8_sliced_k16_f16x8_pack_dbuf, a, b, "(f16x8pack+t8x8+k16+dbuf)", c)
    if args.wmma or args.wmma_all:
        run_benchmark(hgemm.hgemm_wmma_m16n16k16_mma4x2, a, b, "(wmma4x2)", c)
        run_benchmark(hgemm.hgemm_wmma_m16n16k16_mma4x2_warp2x4, a, b, "(wmma4x2+warp2x4)", c)
        for st in (2, 3, 4):
            run_benchmark(hgemm.hgemm_wmma_m16n16k16_mma4x2_warp2x4_stages_dsmem, a, b,
                          f"(wmma4x2+warp2x4+stage{st}+dsmem+swizzle<block>)", c, stages=st, swizzle=True)
    if args.wmma_all:
        run_benchmark(hgemm.hgemm_wmma_m16n16k16_mma4x2_warp4x4_stages_dsmem, a, b,
                      "(wmma4x2+warp4x4+stage2+dsmem+swizzle<block>)", c, stages=2, swizzle=True)
        run_benchmark(hgemm.hgemm_wmma_m16n16k16_mma4x4_warp4x4_stages_dsmem, a, b,
                      "(wmma4x4+warp4x4+stage2+dsmem+swizzle<block>)", c, stages=2, swizzle=True)
    if args.mma_all:
        run_benchmark(hgemm.hgemm_mma_m16n8k16_mma2x4_warp4x4, a, b, "(mma2x4+warp4x4)", c)
        run_benchmark(getattr(hgemm, P + "mma2x4_warp4x4_stages"), a, b, "(mma2x4+warp4x4+stage2+swizzle<block>)", c,
                      stages=2, swizzle=True)
    if args.mma or args.mma_all:
        for st in (2, 3, 4):
            run_benchmark(getattr(hgemm, P + "mma2x4_warp4x4_stages_dsmem"), a, b,
                          f"(mma2x4+warp4x4+stage{st}+dsmem+swizzle<block>)", c, stages=st, swizzle=True)
            run_benchmark(getattr(hgemm, P + "mma2x4_warp4x4x2_stages_dsmem"), a, b,
                          f"(mma2x4+warp4x4x2+stage{st}+dsmem+swizzle<block>)", c, stages=st, swizzle=True)
        run_benchmark(getattr(hgemm, P + "mma2x4_warp4x4x2_stages_dsmem_swizzle"), a, b,
                      "(mma2x4+warp4x4x2+stage2+dsmem+swizzle<smem+block>)", c, stages=2, swizzle=True)
    if args.mma_all:
        run_benchmark(getattr(hgemm, P + "mma2x4_warp4x4x2_stages_dsmem_x4"), a, b,
                      "(mma2x4+warp4x4x2+stage2+dsmem+x4+swizzle<block>)", c, stages=2, swizzle=True)
        run_benchmark(getattr(hgemm, P + "mma2x4_warp4x4x2_stages_dsmem_rr"), a, b,
                      "(mma2x4+warp4x4x2+stage2+dsmem+rr+swizzle<block>)", c, stages=2, swizzle=True)
    if args.mma_tn:
        run_benchmark(getattr(hgemm, P + "mma2x4_warp4x4_stages_dsmem_tn"), a, bc,
                      "tn(mma2x4+warp4x4+stage2+dsmem+swizzle<block>)", c, stages=2, swizzle=True)
        run_benchmark(getattr(hgemm, P + "mma2x4_warp4x4x2_stages_dsmem_tn_swizzle_x4"), a, bc,
                      "tn(mma2x4+warp4x4x2+stage2+dsmem+swizzle<smem+block>)", c, stages=2, swizzle=True)
    if args.cute_tn:
        for st in (2, 3, 4):
            run_benchmark(hgemm.hgemm_mma_stages_block_swizzle_tn_cute, a, bc,
                          f"tn(cute+stage{st}+swizzle<smem+block>)", c, stages=st, swizzle=True)
    if not args.no_cublas:
        run_benchmark(hgemm.hgemm_cublas_tensor_op_nn, a, b, "(cublas)", c)
        run_benchmark(hgemm.hgemm_cublas_tensor_op_tn, a, bc, "tn(cublas)", c)
    if args.torch:
        run_benchmark(partial(torch.matmul, out=c), a, b, "(torch)")
