#!/usr/bin/env python3
"""flash_attn_bench.py — the reference's attention bench driver (kernels/flash-attn/flash_attn_mma.py) on top
of the drop-in module `flash_attn_lib`: same call convention (`lib.<entry>(q, k, v, o, stages)`), warm-up 1 /
iters 5 wall-clock timing (:345-377), TFLOPS from get_mha_tflops (:241-278), `--check` =
torch.allclose(ref, out, atol=1e-2) + max/min/mean diff line (:465-494) against SDPA (the `flash_attn` pip
package is not installed here), V handed over transposed for the *_swizzle_qkv share/tiling_qk entries (:441).

    PYTHONPATH=leetcuda_amd python tools/flash_attn_bench.py --B 4 --H 32 --N 4096 --D 128 --check
"""
import argparse
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "leetcuda_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from leetcuda_amd import capi  # noqa: E402
from leetcuda_amd.host import get_mha_tflops  # noqa: E402

ap = argparse.ArgumentParser()
for f, d in (("B", 4), ("H", 32), ("N", 4096), ("D", 128)):
    ap.add_argument(f"--{f}", type=int, default=d)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--warmup", "--w", type=int, default=1)
ap.add_argument("--iters", "--i", type=int, default=5)
ap.add_argument("--check", action="store_true")
ap.add_argument("--flops-mm", action="store_true")
args = ap.parse_args()
import flash_attn_lib as lib  # noqa: E402

torch.manual_seed(args.seed)
B, H, N, D = args.B, args.H, args.N, args.D
q = torch.randn((B, H, N, D), dtype=torch.half, device="cuda")
k = torch.randn((B, H, N, D), dtype=torch.half, device="cuda")
v = torch.randn((B, H, N, D), dtype=torch.half, device="cuda")
o = torch.zeros_like(q)
tv = v.transpose(-2, -1).contiguous()
ref = F.scaled_dot_product_attention(q, k, v) if args.check else None
print("-" * 150)
print(f"B={B}, H={H}, N={N}, D={D}, Warmup: {args.warmup}, Iters: {args.iters}".center(150))
print("-" * 150)
best = -1.0
for name, fam, vt, acc, d2, d1, nargs in capi.attn_entries():
    for stages in ((2,) if nargs == 4 else (1, 2)):
        if D > (d2 if stages > 1 else d1):
            continue   # MAX_HEADDIM_CFG of the reference bench (:504-576): entry not defined for this head dim
        fn = getattr(lib, name)
        vv = tv if vt else v
        call = (lambda: fn(q, k, vv, o)) if nargs == 4 else (lambda: fn(q, k, vv, o, stages))
        o.zero_()
        for _ in range(args.warmup):
            call()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(args.iters):
            call()
        torch.cuda.synchronize()
        mean = (time.time() - t0) / args.iters
        tf = get_mha_tflops(B, H, N, D, mean, only_matmul=args.flops_mm)
        tag = name.replace("flash_attn_mma_stages_", "mma(").replace("_", "+") + f"+stage{stages})"
        imp = f"(+{(tf - best) / best * 100:.2f}%)" if 0 < best < tf else ""
        best = max(best, tf)
        vals = [round(x, 6) for x in o.flatten()[:3].float().tolist()]
        print(f"{tag:>70}: {vals}, time:{mean * 1000:.6f}ms, TFLOPS:{tf:<7.2f}{imp}")
        if args.check:
            diff = (ref - o).abs()
            ok = torch.allclose(ref, o, atol=1e-2)
            print(f"{'out_sdpa vs ' + tag:>70}, all close: {str(ok):<6}, max diff: {diff.max().item():.6f}, "
                  f"min diff: {diff.min().item():.6f}, mean diff: {diff.mean().item():.6f}")
