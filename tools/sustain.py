#!/usr/bin/env python3
"""Back-to-back launches of one hot kernel for a few seconds per spec (the workload of tools/power_watch.sh; join the
sampler log with tools/power_join.py).  usage: sustain.py [--seconds S] spec...
   spec = hgemm | vendor | attn | attn8k | d512 | d256 (:bf16, :d512=K = lc_tune_set "attn_d512") | fp8 (16384^3; :k64 = the K = 64 kernel, :mx = lc_gemm_mxfp8 with block scales 2^-2 .. 2^2), optionally :nn (B stored [K][N]), :zero (zero-filled inputs), :uniform, :var=auto|w4c|w4x|pingpong2, :nopersist (hgemm: one tile per workgroup), :stg=N (hgemm K-loop stagger; stg=0 = off, no option = the default), :abl=N (hgemm only:
          lc_tune_set "w4_abl", LC_DIAG library — ablated kernels compute WRONG results), :nw=N (attention kernel choice)"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from leetcuda_amd import capi, host  # noqa: E402

args = sys.argv[1:]
secs = 2.5
if args and args[0] == "--seconds":
    secs = float(args[1])
    args = args[2:]
capi.load()
vendor_ready = False
VAR = {"auto": capi.HGEMM_AUTO, "w4c": capi.HGEMM_MFMA256W4C, "w4x": capi.HGEMM_MFMA256W4X, "w4y": capi.HGEMM_MFMA256W4Y, "pingpong2": capi.HGEMM_MFMA256P2}


def fill(t, opts):
    if "zero" in opts:
        t.zero_()
    elif "uniform" in opts:
        t.uniform_(-1, 1)
    return t


for spec in args:
    what, *opts = spec.split(":")
    abl = next((int(o[4:]) for o in opts if o.startswith("abl=")), 0)
    nw = next((int(o[3:]) for o in opts if o.startswith("nw=")), 0)
    var = next((o[4:] for o in opts if o.startswith("var=")), "auto")
    capi.tune("w4y_sched", next((int(o[6:]) for o in opts if o.startswith("sched=")), 1))
    capi.tune("hgemm_persist", 0 if "nopersist" in opts else 1)
    capi.tune("hgemm_stagger", next((int(o[4:], 0) or 1 << 27 for o in opts if o.startswith("stg=")), 0))   # cx | cm << 4 | cn << 8 | step << 12 | mask << 20
    if what in ("hgemm", "vendor"):
        n = 8192
        a = fill(torch.randn(n, n, dtype=torch.half, device="cuda"), opts)
        lay = capi.LAYOUT_NN if "nn" in opts else capi.LAYOUT_TN
        b = fill(torch.randn(n, n, dtype=torch.half, device="cuda"), opts)
        if lay == capi.LAYOUT_TN:
            b = host.as_col_major(b)
        c = torch.empty(n, n, dtype=torch.half, device="cuda")
        flops = 2.0 * n ** 3
        if what == "vendor":
            if not vendor_ready:
                capi.vendor_init()
                vendor_ready = True
            step = lambda: capi.hgemm_vendor(a, b, c, layout=lay)  # noqa: E731
        else:
            if abl:
                capi.tune("w4_abl", abl)
                var = "w4c"
            step = lambda: capi.hgemm(a, b, c, layout=lay, variant=VAR[var], swizzle_stride=2048)  # noqa: E731
    elif what in ("attn", "attn8k"):
        B, H, N, D = (4, 32, 4096, 128) if what == "attn" else (4, 32, 8192, 128)
        q, k, v, o, _ = host.get_qkvo(B, H, N, D, seed=0)
        for t in (q, k, v):
            fill(t, opts)
        capi.tune("attn_nw", nw)
        step = lambda: capi.attn_fwd(q, k, v, o)  # noqa: E731
        flops = host.mha_matmul_flops(B, H, N, D)
    elif what == "d512":
        q = fill(torch.randn(1, 48, 8192, 512, device="cuda").half(), opts)
        k = fill(torch.randn(1, 48, 8192, 512, device="cuda").half(), opts)
        v = fill(torch.randn(1, 48, 8192, 512, device="cuda").half(), opts)
        o = torch.zeros_like(q)
        step = lambda: capi.attn_call("flash_attn_mma_stages_split_q_tiling_qkv", q, k, v, o, 2)  # noqa: E731
        flops = host.mha_matmul_flops(1, 48, 8192, 512)
    elif what == "d256":
        dt = torch.bfloat16 if "bf16" in opts else torch.half
        q = fill(torch.randn(1, 48, 8192, 256, device="cuda").to(dt), opts)
        k = fill(torch.randn(1, 48, 8192, 256, device="cuda").to(dt), opts)
        v = fill(torch.randn(1, 48, 8192, 256, device="cuda").to(dt), opts)
        o = torch.zeros_like(q)
        capi.tune("attn_d512", next((int(x[5:]) for x in opts if x.startswith("d512=")), 0))
        step = (lambda: capi.attn_fwd_bf16(q, k, v, o)) if "bf16" in opts else \
            (lambda: capi.attn_call("flash_attn_mma_stages_split_q_tiling_qkv", q, k, v, o, 2))  # noqa: E731
        flops = host.mha_matmul_flops(1, 48, 8192, 256)
    elif what == "fp8":
        n = 16384
        a8 = fill(torch.randn(n, n, device="cuda"), opts).to(torch.float8_e4m3fn)
        b8 = fill(torch.randn(n, n, device="cuda"), opts).to(torch.float8_e4m3fn)
        c8 = torch.empty(n, n, dtype=torch.half, device="cuda")
        capi.tune("fp8_mx", 1 if "k64" in opts else 3)
        if "mx" in opts:
            pa = capi.mxfp8_pack_scales(torch.randint(125, 130, (n, n // 32), device="cuda", dtype=torch.uint8))
            pb = capi.mxfp8_pack_scales(torch.randint(125, 130, (n, n // 32), device="cuda", dtype=torch.uint8))
            step = lambda: capi.gemm_mxfp8(a8, pa, b8, pb, c8, alpha=1 / 64, swizzle_stride=2048)  # noqa: E731
        else:
            step = lambda: capi.gemm_fp8(a8, b8, c8, alpha=1 / 16, swizzle_stride=2048)  # noqa: E731
        flops = 2.0 * n ** 3
    else:
        raise SystemExit(f"unknown spec {spec}")
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t_start = time.time()
    n_l = 0
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t_start < secs:
        for _ in range(100):
            step()
        n_l += 100
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print(f"RUN {spec:24s} {flops * n_l / ms * 1e-9:7.1f} TFLOP/s(-equivalent)  t0={t_start:.3f} t1={time.time():.3f}", flush=True)
    if abl:
        capi.tune("w4_abl", 0)
    if nw:
        capi.tune("attn_nw", 0)
    capi.tune("attn_d512", 0)
    capi.tune("fp8_mx", 3)
