#!/usr/bin/env python3
"""Generate tests/golden/* by RUNNING the reference's own Python callables in this container.

The reference CUDA kernels cannot run here (no nvcc / NVIDIA GPU), and it ships no golden vectors
(SURVEY.md §8c).  What can run is the host-side Python of its bench drivers and the in-script CPU-capable
baselines it compares against.  This script lifts those function definitions verbatim BY AST from
/root/reference (nothing is copied into the repo), executes them on seeded CPU tensors and stores
inputs + outputs as small fixtures:

  kernels/hgemm/tools/utils.py      as_col_major
  kernels/hgemm/hgemm.py            make_block_swizzle_stride        (+ the TFLOPS formula, :282)
  kernels/flash-attn/flash_attn_mma.py   get_mha_tflops, unfused_standard_attn
  torch.matmul / F.scaled_dot_product_attention on CPU  (hgemm.py:1088, flash_attn_mma.py:455-462)

Run only where /root/reference exists:  python tools/make_golden.py
"""
import ast
import json
import math
import sys
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

REF = Path("/root/reference/kernels")
OUT = Path(__file__).resolve().parent.parent / "tests" / "golden"


def lift(path: Path, names):
    """exec only the requested top-level function definitions of a reference file."""
    tree = ast.parse(path.read_text())
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    missing = set(names) - {n.name for n in body}
    if missing:
        raise SystemExit(f"{path}: missing {missing}")
    for fn in body:
        fn.decorator_list = [d for d in fn.decorator_list
                             if not (isinstance(d, ast.Attribute) and d.attr == "no_grad")]
    mod = ast.Module(body=body, type_ignores=[])
    env = {"torch": torch, "math": math, "F": F, "Optional": __import__("typing").Optional}
    exec(compile(mod, str(path), "exec"), env)
    return [env[n] for n in names]


def main():
    if not REF.exists():
        sys.exit("reference not mounted; fixtures are committed, nothing to do")
    OUT.mkdir(parents=True, exist_ok=True)
    (as_col_major,) = lift(REF / "hgemm/tools/utils.py", ["as_col_major"])
    (make_block_swizzle_stride,) = lift(REF / "hgemm/hgemm.py", ["make_block_swizzle_stride"])
    get_mha_tflops, unfused_standard_attn = lift(REF / "flash-attn/flash_attn_mma.py",
                                                 ["get_mha_tflops", "unfused_standard_attn"])

    # ---- host bookkeeping -----------------------------------------------------------------------
    host = {"swizzle_stride": [], "mha_tflops": [], "source": "xlite-dev/LeetCUDA @ 2025-07-18"}
    for N in (256, 512, 1024, 2048, 4096, 4352, 8192, 14848, 15104, 16384):
        for K in (256, 2048, 8192, 8448, 16384):
            for f in (None, 0.5, 0.25, 0.125, 0.0625):
                host["swizzle_stride"].append([N, K, f, make_block_swizzle_stride(N, K, f)])
    for (B, H, N, D) in ((1, 8, 8192, 64), (4, 32, 4096, 128), (32, 32, 8192, 128), (1, 48, 8192, 512),
                         (2, 3, 256, 96)):
        for om in (False, True):
            host["mha_tflops"].append([B, H, N, D, 1.0, om, get_mha_tflops(B, H, N, D, 1.0, om)])
    (OUT / "host_helpers.json").write_text(json.dumps(host, indent=1))

    # ---- as_col_major ---------------------------------------------------------------------------
    torch.manual_seed(1234)
    x = torch.randn(24, 40, dtype=torch.half)
    np.savez_compressed(OUT / "as_col_major.npz", x=x.numpy().view(np.uint16),
                        y=as_col_major(x).numpy().view(np.uint16))

    # ---- HGEMM: the reference's torch baseline on CPU tensors ------------------------------------
    cases = {}
    for i, (M, N, K) in enumerate(((64, 64, 64), (128, 256, 96), (256, 256, 128), (96, 80, 200))):
        torch.manual_seed(100 + i)
        a = torch.randn(M, K, dtype=torch.half)
        b = torch.randn(K, N, dtype=torch.half)
        c16 = torch.matmul(a, b)                       # partial(torch.matmul, out=c), fp16 CPU
        c32 = torch.matmul(a.float(), b.float())       # fp32 math on the same rounded inputs
        c64 = torch.matmul(a.double(), b.double())
        cases[f"a{i}"] = a.numpy().view(np.uint16)
        cases[f"b{i}"] = b.numpy().view(np.uint16)
        cases[f"bcol{i}"] = as_col_major(b).numpy().view(np.uint16)
        cases[f"c16_{i}"] = c16.numpy().view(np.uint16)
        cases[f"c32_{i}"] = c32.numpy()
        cases[f"c64_{i}"] = c64.float().numpy()   # fp64 result, stored as fp32
    np.savez_compressed(OUT / "hgemm_small.npz", **cases)

    # ---- attention: unfused_standard_attn + SDPA on CPU tensors ----------------------------------
    cases = {}
    for i, (B, H, N, D) in enumerate(((1, 2, 128, 64), (1, 1, 256, 128), (2, 1, 64, 32), (1, 1, 128, 96))):
        torch.manual_seed(200 + i)
        q = torch.randn(B, H, N, D, dtype=torch.half)
        k = torch.randn(B, H, N, D, dtype=torch.half)
        v = torch.randn(B, H, N, D, dtype=torch.half)
        o32 = unfused_standard_attn(q.float(), k.float(), v.float())
        o64 = unfused_standard_attn(q.double(), k.double(), v.double())
        o16 = unfused_standard_attn(q, k, v)
        sd = F.scaled_dot_product_attention(q, k, v)
        for nm, t in (("q", q), ("k", k), ("v", v), ("o16", o16), ("sdpa16", sd)):
            cases[f"{nm}{i}"] = t.numpy().view(np.uint16)
        cases[f"o32_{i}"] = o32.numpy()
        cases[f"o64_{i}"] = o64.float().numpy()   # fp64 result, stored as fp32
    np.savez_compressed(OUT / "attn_small.npz", **cases)
    print("wrote", sorted(p.name for p in OUT.iterdir()))


if __name__ == "__main__":
    main()
