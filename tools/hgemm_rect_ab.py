#!/usr/bin/env python3
"""A/B of LC_HGEMM_AUTO on RECTANGULAR shapes (the launch rules were fitted on the reference's square sweep): per (M, N, K) and layout
the auto launch against every forced alternative that is legal for the shape — the mid-size kernel's tiles, the 256-tile kernel, the
128-tile kernel — and hipBLASLt; `seconds` sustained per cell in three rounds that rotate the candidates.

    tools/hgemm_rect_ab.py [MxNxK,MxNxK,...] [seconds] [tn|nn|both]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from leetcuda_amd import capi, host  # noqa: E402

capi.load()
capi.vendor_init()
DEFAULT = ("1024x4096x4096,2048x4096x4096,4096x1024x4096,4096x4096x1024,1024x1024x8192,2048x2048x8192,512x8192x2048,8192x512x2048,"
           "1536x3072x3072,3072x1536x3072,2304x3072x768,1024x11008x4096,2048x11008x4096,1024x4096x11008,4096x2048x2048,2048x8192x1024,"
           "1280x5120x5120,2560x1280x5120,768x3072x3072,3072x768x3072,1792x7168x1792,1024x14336x4096,2048x14336x4096,256x16384x4096")
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


worst = []
for (M, N, K) in shapes:
    a = torch.randn(M, K, dtype=torch.half, device="cuda")
    b = torch.randn(K, N, dtype=torch.half, device="cuda")
    c = torch.empty(M, N, dtype=torch.half, device="cuda")
    fl = 2.0 * M * N * K
    st = host.make_block_swizzle_stride(N, K)
    for lname, lay in (("tn", capi.LAYOUT_TN), ("nn", capi.LAYOUT_NN)):
        if LAYS not in ("both", lname):
            continue
        b2 = host.as_col_major(b) if lay == capi.LAYOUT_TN else b
        cands = {}

        def mk(mid, ns, var, sk=1):
            def f():
                capi.tune("hgemm_mid", mid)
                capi.tune("hgemm_mid_ns", ns)
                capi.tune("hgemm_mid_splitk", sk)
                capi.hgemm(a, b2, c, layout=lay, variant=var, swizzle_stride=st)
            return f
        cands["auto"] = mk(0, 0, capi.HGEMM_AUTO, 0)
        for tmw in (2, 3, 1):
            for w in (2, 3):
                if N % (64 * w) or M % (64 * tmw) or (lname == "nn" and w == 3):
                    continue
                for ns in (2, 3):
                    cands[f"mid{tmw}{w}x{ns}"] = mk(10 * tmw + w, ns, capi.HGEMM_MID)
                if w == 2 and tmw <= 2 and (M // (64 * tmw)) * (N // 128) <= 128:
                    for sk in (2, 4, 8):
                        if K // 64 >= 4 * sk:
                            cands[f"mid{tmw}{w}sk{sk}"] = mk(10 * tmw + w, 3, capi.HGEMM_MID, sk)
        if M % 256 == 0 and N % 256 == 0:
            cands["w4y"] = mk(1, 0, capi.HGEMM_MFMA256W4Y)
        if M % 128 == 0 and N % 128 == 0:
            cands["m128"] = mk(1, 0, capi.HGEMM_MFMA128)
        cands["hipBLASLt"] = lambda: capi.hgemm_vendor(a, b2, c, lay)
        for f in cands.values():
            burst(f, 3)
        per = burst(cands["auto"], 10) / 10
        cnt = max(10, int(SECONDS / 3 / per))
        t = {k: 0.0 for k in cands}
        keys = list(cands)
        for r in range(3):
            for k in keys[r:] + keys[:r]:
                t[k] += burst(cands[k], cnt)
        capi.tune("hgemm_mid", 0)
        capi.tune("hgemm_mid_ns", 0)
        capi.tune("hgemm_mid_splitk", 0)
        rate = {k: fl * cnt * 3 / v * 1e-12 for k, v in t.items()}
        best = max((k for k in rate if k not in ("auto", "hipBLASLt")), key=lambda k: rate[k], default="auto")
        worst.append((rate["auto"] / rate[best], rate["auto"] / rate["hipBLASLt"], M, N, K, lname, best))
        print(f"{M}x{N}x{K} {lname} auto={capi.hgemm_kernel_name(M, N, K, lay)} {rate['auto']:6.1f} | best forced {best} {rate[best]:6.1f} "
              f"(auto / best {rate['auto'] / rate[best]:.3f}) | hipBLASLt {rate['hipBLASLt']:6.1f} (auto / vendor {rate['auto'] / rate['hipBLASLt']:.3f}) | "
              + " ".join(f"{k} {v:.0f}" for k, v in rate.items() if k not in ("auto", "hipBLASLt")), flush=True)
worst.sort()
print("auto furthest behind its best forced alternative:", [(round(r, 3), f"{m}x{n}x{k} {l} {b}") for r, _, m, n, k, l, b in worst[:8]])
worst.sort(key=lambda w: w[1])
print("auto furthest behind hipBLASLt:", [(round(v, 3), f"{m}x{n}x{k} {l}") for _, v, m, n, k, l, _ in worst[:8]])
capi.vendor_destroy()
