#!/usr/bin/env python3
"""A/B of the two edge kernels on shapes no tiled kernel divides: hgemm_edge_kernel (16-byte chunks, LC_HGEMM_EDGE — what LC_HGEMM_AUTO
runs for K % 8 == 0, NN: N % 8 == 0), LC_HGEMM_RAGGED (K % 32 == 0, N % 8 == 0: interior on the tiled kernels + border on the edge kernel) against hgemm_generic_kernel (element-wise staging, LC_HGEMM_GENERIC) and hipBLASLt; `seconds`
sustained per cell in three rounds that rotate the candidates.

    tools/hgemm_edge_ab.py [MxNxK,MxNxK,...] [seconds] [tn|nn|both]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from leetcuda_amd import capi, host  # noqa: E402

capi.load()
capi.vendor_init()
DEFAULT = ("2880x2880x2880,8192x8256x4096,1000x3000x4096,4100x4088x4104,8192x136x8200,8200x8200x8200,5000x5000x5000,1000x1000x1000,"
           "8192x8224x8192,8192x8192x8200,12808x12808x4096,333x4096x4096,5000x5000x4096,4100x4104x4096,3000x3000x3008,2000x2000x2048,"
           "1500x1504x1536,8200x8200x8192,4000x11008x4096,777x50264x4096")
shapes = [tuple(int(v) for v in s.split("x")) for s in (sys.argv[1] if len(sys.argv) > 1 else DEFAULT).split(",")]
SECONDS = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
LAYS = sys.argv[3] if len(sys.argv) > 3 else "both"


def burst(step, n):
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n):
        step()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) * 1e-3


for (M, N, K) in shapes:
    a = torch.randn(M, K, dtype=torch.half, device="cuda")
    b = torch.randn(K, N, dtype=torch.half, device="cuda")
    c = torch.empty(M, N, dtype=torch.half, device="cuda")
    fl = 2.0 * M * N * K
    for lname, lay in (("tn", capi.LAYOUT_TN), ("nn", capi.LAYOUT_NN)):
        if LAYS not in ("both", lname):
            continue
        b2 = host.as_col_major(b) if lay == capi.LAYOUT_TN else b
        cands = {"auto": lambda: capi.hgemm(a, b2, c, layout=lay, variant=capi.HGEMM_AUTO, swizzle_stride=host.make_block_swizzle_stride(N, K)),
                 "generic": lambda: capi.hgemm(a, b2, c, layout=lay, variant=capi.HGEMM_GENERIC)}
        try:
            capi.hgemm_kernel_name(M, N, K, lay, capi.HGEMM_EDGE)
            cands["edge"] = lambda: capi.hgemm(a, b2, c, layout=lay, variant=capi.HGEMM_EDGE)
        except capi.LcError:
            pass
        try:
            capi.hgemm_kernel_name(M, N, K, lay, capi.HGEMM_RAGGED)
            cands["ragged"] = lambda: capi.hgemm(a, b2, c, layout=lay, variant=capi.HGEMM_RAGGED, swizzle_stride=host.make_block_swizzle_stride(N, K))

            def mk(knob):
                def f():
                    capi.tune("hgemm_ragged_fork", knob)
                    capi.hgemm(a, b2, c, layout=lay, variant=capi.HGEMM_RAGGED, swizzle_stride=host.make_block_swizzle_stride(N, K))
                    capi.tune("hgemm_ragged_fork", 0)
                return f
            if "+" in capi.hgemm_kernel_name(M, N, K, lay, capi.HGEMM_RAGGED):
                cands["ragged_seq"] = mk(1)       # border behind the interior on the caller's stream
                cands["ragged_fork"] = mk(2)      # border on the side stream
            else:
                def mkt(tile):
                    def f():
                        capi.tune("hgemm_ragged_tile", tile)
                        capi.hgemm(a, b2, c, layout=lay, variant=capi.HGEMM_RAGGED, swizzle_stride=host.make_block_swizzle_stride(N, K))
                        capi.tune("hgemm_ragged_tile", 0)
                    return f
                for tile in ((12, 22, 32) if lname == "nn" else (12, 22, 23, 33)):
                    cands[f"t{tile}"] = mkt(tile)

                def mks(ks):
                    def f():
                        capi.tune("hgemm_mid_splitk", ks)
                        capi.hgemm(a, b2, c, layout=lay, variant=capi.HGEMM_RAGGED, swizzle_stride=host.make_block_swizzle_stride(N, K))
                        capi.tune("hgemm_mid_splitk", 0)
                    return f
                if K // 64 >= 8 and (-(-M // 128)) * (-(-N // 128)) <= 128:
                    for ks in (1, 2, 4, 8):
                        if K // 64 >= 2 * ks:
                            cands[f"sk{ks}"] = mks(ks)
        except capi.LcError:
            pass
        try:
            capi.hgemm_kernel_name(M, N, K, lay, capi.HGEMM_KPAD)
            cands["kpad"] = lambda: capi.hgemm(a, b2, c, layout=lay, variant=capi.HGEMM_KPAD, swizzle_stride=host.make_block_swizzle_stride(N, K))
        except capi.LcError:
            pass
        cands["hipBLASLt"] = lambda: capi.hgemm_vendor(a, b2, c, lay)
        for f in cands.values():
            burst(f, 2)
        t = {k: 0.0 for k in cands}
        n = {k: 0 for k in cands}
        keys = list(cands)
        for r in range(3):
            for k in keys[r:] + keys[:r]:
                per = burst(cands[k], 2) / 2
                cnt = max(2, int(SECONDS / 3 / per))
                t[k] += burst(cands[k], cnt)
                n[k] += cnt
        rate = {k: fl * n[k] / t[k] * 1e-12 for k in cands}
        print(f"{M}x{N}x{K} {lname} auto={capi.hgemm_kernel_name(M, N, K, lay)} " + " ".join(f"{k} {v:7.1f}" for k, v in rate.items())
              + (f" | edge / generic {rate['edge'] / rate['generic']:.2f}, edge / vendor {rate['edge'] / rate['hipBLASLt']:.3f}" if "edge" in rate else "")
              + (f", ragged / edge {rate['ragged'] / rate['edge']:.2f}" if "ragged" in rate and "edge" in rate else "")
              + (f", kpad / edge {rate['kpad'] / rate['edge']:.2f}" if "kpad" in rate and "edge" in rate else "")
              + f", auto / vendor {rate['auto'] / rate['hipBLASLt']:.3f}",
              flush=True)
capi.vendor_destroy()
