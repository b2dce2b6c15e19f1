#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload hgemm|attn|attn_sharded]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): achieved fp16 TFLOPS vs MI355X MFMA peak — HGEMM 8192^3; FA-2 fwd S=4096 D=128.
A "step" is one pass of the hot path over one batch of synthetic input already resident in HBM:

  hgemm (default)   one C[8192,8192] = A·B fp16 GEMM per rank (BASELINE config 2), TN storage for B
                    (the layout of the reference's fastest kernel, hgemm_mma_stage_tn_cute.cu).
                    N > 1: N independent replicas (HGEMM does not shard in north_star) -> "weak".
  attn              FlashAttention-2 forward B=4,H=32,S=4096,D=128 (config 3); N > 1: the 128 (batch,head)
                    problems are split across ranks -> "strong".
  attn_sharded      config 4: B=32,H=32,S=8192,D=128 batch-sharded over N ranks -> "strong".

The default run reports HGEMM as `value` and carries the config-3 attention numbers in "attention".
No data-path collective exists: the only collectives are the barrier bracketing the timed region and
the gather of per-rank timings (RCCL when N > 1).  W warm-up steps, then EXACTLY K timed steps between
barrier + torch.cuda.synchronize() on both sides; time = MAX over ranks; rank 0 prints ONE JSON line.

"roofline": achieved = algorithmic FLOPs per launch / average launch duration from HIP events recorded on
the launch stream (lc_hgemm_time / lc_attn_time); peak = 2500 TFLOP/s dense fp16 MFMA.
"cpu_baseline": the reference benches' own CPU-capable baseline callables (torch.matmul, hgemm.py:1088;
F.scaled_dot_product_attention, flash_attn_mma.py:455-462) timed on this box's host cores, rank 0, N=1,
on a bounded sample — a reported baseline, not the target.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

from leetcuda_amd import capi, host  # noqa: E402
from leetcuda_amd import dist as lcd  # noqa: E402

PEAK = host.MI355X_FP16_DENSE_PEAK_TFLOPS
PREWARM = 10  # untimed launches before the W warm-up steps (DVFS settles; documented in DESIGN.md §6)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="hgemm", choices=["hgemm", "attn", "attn_sharded"])
    ap.add_argument("--layout", default="tn", choices=["tn", "nn"])
    ap.add_argument("--variant", default="auto", choices=["auto", "mfma256", "pingpong", "pingpong2", "pingpong3", "w4", "w4s", "w4b", "w4c", "generic"])
    ap.add_argument("--mnk", type=int, default=8192)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-attention", action="store_true", help="skip the secondary attention measurement")
    ap.add_argument("--sweep", action="store_true", help="also print a per-variant table to stderr")
    return ap.parse_args()


VARIANT = {"auto": capi.HGEMM_AUTO, "mfma256": capi.HGEMM_MFMA256, "pingpong": capi.HGEMM_MFMA256P,
           "pingpong2": capi.HGEMM_MFMA256P2, "pingpong3": capi.HGEMM_MFMA256P3, "w4": capi.HGEMM_MFMA256W4, "w4s": capi.HGEMM_MFMA256W4S, "w4b": capi.HGEMM_MFMA256W4B, "w4c": capi.HGEMM_MFMA256W4C, "generic": capi.HGEMM_GENERIC}
AUTO_KERNEL = "w4c"   # what LC_HGEMM_AUTO resolves to (lc_abi.hip: g_tune_hgemm_auto)


def pmc_key_hgemm(variant: str, layout: str) -> str:
    v = AUTO_KERNEL if variant == "auto" else variant
    nn = 'true' if layout == 'nn' else 'false'
    if v in ("w4b", "w4c"):
        return f"hgemm_w4b_kernel<{nn},{'true' if v == 'w4c' else 'false'}>"
    if v == "w4":
        return f"hgemm_w4_kernel<{nn},0>"
    return f"hgemm_{v}_kernel<{nn}>"


def timed_region(w, step, steps, warmup):
    """W untimed + exactly K timed steps, barrier+sync on both sides; returns local seconds."""
    for _ in range(PREWARM):   # setup: clocks / code objects / allocator, not part of W or K
        step()
    for _ in range(warmup):
        step()
    lcd.barrier(w)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    lcd.barrier(w)
    return time.perf_counter() - t0


def pmc_traffic(kernel: str):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes
    (profiles/latest_pmc.json, written by tools/summarize_prof.py: FETCH_SIZE x2 + WRITE_SIZE), or None."""
    try:
        return json.loads((ROOT / "profiles" / "latest_pmc.json").read_text())[kernel]["hbm_bytes_per_launch"]
    except Exception:
        return None


# ---------------------------------------------------------------------------------------------------
def bench_hgemm(w, args):
    n = args.mnk
    lay = capi.LAYOUT_TN if args.layout == "tn" else capi.LAYOUT_NN
    var = VARIANT[args.variant]
    torch.manual_seed(0 + w.rank)
    a = torch.randn((n, n), dtype=torch.half, device="cuda")     # hgemm.py:444-446
    b = torch.randn((n, n), dtype=torch.half, device="cuda")
    c = torch.zeros((n, n), dtype=torch.half, device="cuda")
    bb = host.as_col_major(b) if lay == capi.LAYOUT_TN else b
    stride = host.make_block_swizzle_stride(n, n)                # reference default: N/4 = 2048 at 8192
    step = lambda: capi.hgemm(a, bb, c, layout=lay, variant=var, stages=2, swizzle_stride=stride)  # noqa: E731
    secs = timed_region(w, step, args.steps, args.warmup)
    secs = lcd.max_over_ranks(w, secs)
    flops = 2.0 * n * n * n
    ms_kernel = capi.hgemm_time(a, bb, c, lay, var, 2, stride, warmup=2, iters=max(10, args.steps))
    ms_kernel = lcd.max_over_ranks(w, ms_kernel)
    res = {
        "value": w.size * flops * args.steps / secs * 1e-12,
        "ms_per_step": secs / args.steps * 1e3,
        "workload": f"HGEMM M=N=K={n} fp16 {args.layout.upper()} (BASELINE config 2), randn inputs, "
                    f"variant={args.variant}, block-swizzle stride {stride}",
        "scaling": "weak",
        "roofline": {"bound": "mfma", "achieved": flops / (ms_kernel * 1e-3) * 1e-12, "peak": PEAK,
                     "unit": "TFLOP/s", "kernel_ms": ms_kernel,
                     "kernel": pmc_key_hgemm(args.variant, args.layout),
                     "algorithmic_flops_per_launch": flops,
                     "algorithmic_bytes_per_launch": 3.0 * n * n * 2,
                     "traffic": pmc_traffic(pmc_key_hgemm(args.variant, args.layout))},
    }
    res["roofline"]["frac"] = res["roofline"]["achieved"] / PEAK
    if args.sweep and w.rank == 0:
        capi.vendor_init()
        for lname, l2 in (("tn", capi.LAYOUT_TN), ("nn", capi.LAYOUT_NN)):
            b2 = host.as_col_major(b) if l2 == capi.LAYOUT_TN else b
            for vn in ("mfma256", "pingpong", "pingpong2", "pingpong3", "w4", "w4s", "w4b", "w4c"):
                for st in (1, 1024, 2048):
                    ms = capi.hgemm_time(a, b2, c, l2, VARIANT[vn], 2, st, warmup=2, iters=20)
                    print(f"[sweep] hgemm {lname} {vn:9s} stride {st:5d}: {ms:.4f} ms  "
                          f"{flops / ms * 1e-9:8.1f} TFLOP/s", file=sys.stderr)
            t0 = torch.cuda.Event(enable_timing=True)
            t1 = torch.cuda.Event(enable_timing=True)
            for _ in range(3):
                capi.hgemm_vendor(a, b2, c, l2)
            t0.record()
            for _ in range(20):
                capi.hgemm_vendor(a, b2, c, l2)
            t1.record()
            torch.cuda.synchronize()
            ms = t0.elapsed_time(t1) / 20
            print(f"[sweep] hgemm {lname} hipBLASLt            : {ms:.4f} ms  {flops / ms * 1e-9:8.1f} TFLOP/s",
                  file=sys.stderr)
            res.setdefault("vendor_tflops", {})[lname] = flops / ms * 1e-9
        capi.vendor_destroy()
        q, k, v, o, _ = host.get_qkvo(4, 32, 4096, 128)
        fl = host.mha_matmul_flops(4, 32, 4096, 128)
        for nw in (32, 16, 8, 4):
            capi.tune("attn_nw", nw)
            ms = capi.attn_time(q, k, v, o, False, capi.ATTN_SPLIT_Q, 2, warmup=2, iters=10)
            print(f"[sweep] attn cfg3 nw={nw}: {ms:.4f} ms  {fl / ms * 1e-9:8.1f} TFLOP/s", file=sys.stderr)
        capi.tune("attn_nw", 0)
        for nn in (8192, 16384):     # config-5 extension: fp8 e4m3 GEMM (TN), alpha = 1/16
            a8 = torch.randn(nn, nn, device="cuda").to(torch.float8_e4m3fn)
            b8 = torch.randn(nn, nn, device="cuda").to(torch.float8_e4m3fn)
            c8 = torch.zeros(nn, nn, dtype=torch.half, device="cuda")
            for _ in range(3):
                capi.gemm_fp8(a8, b8, c8, alpha=1 / 16, swizzle_stride=2048)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                capi.gemm_fp8(a8, b8, c8, alpha=1 / 16, swizzle_stride=2048)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            print(f"[sweep] gemm fp8 e4m3 {nn}^3: {ms:.4f} ms  {2.0 * nn ** 3 / ms * 1e-9:8.1f} TFLOP/s", file=sys.stderr)
            res.setdefault("fp8_tflops", {})[str(nn)] = 2.0 * nn ** 3 / ms * 1e-9
            del a8, b8, c8
    return res


def bench_attn(w, args, sharded_cfg4=False, steps=None, warmup=None):
    steps = steps or args.steps
    warmup = args.warmup if warmup is None else warmup
    B, H, N, D = (32, 32, 8192, 128) if sharded_cfg4 else (4, 32, 4096, 128)
    b_loc, h_loc, _ = host.attn_shard(B, H, w.size, w.rank)
    torch.manual_seed(0 + w.rank)
    q, k, v, o, _ = host.get_qkvo(b_loc, h_loc, N, D)            # flash_attn_mma.py:417-435
    fam = capi.ATTN_SHARED_QKV if sharded_cfg4 else capi.ATTN_SPLIT_Q
    step = lambda: capi.attn_fwd(q, k, v, o, family=fam, stages=2)  # noqa: E731
    secs = timed_region(w, step, steps, warmup)
    secs = lcd.max_over_ranks(w, secs)
    flops_total = host.mha_matmul_flops(B, H, N, D)               # whole job, all ranks
    flops_local = host.mha_matmul_flops(b_loc, h_loc, N, D)
    ms_kernel = capi.attn_time(q, k, v, o, False, fam, 2, warmup=1, iters=max(5, steps))
    ms_kernel = lcd.max_over_ranks(w, ms_kernel)
    ach = flops_local / (ms_kernel * 1e-3) * 1e-12
    return {
        "value": flops_total * steps / secs * 1e-12,
        "ms_per_step": secs / steps * 1e3,
        "tflops_reference_formula": host.get_mha_tflops(B, H, N, D, secs / steps),
        "workload": f"FlashAttention-2 fwd B={B} H={H} S={N} D={D} fp16 "
                    f"({'config 4, shared-QKV entry, batch-sharded' if sharded_cfg4 else 'config 3, split-Q entry'}), "
                    f"randn inputs, {b_loc}x{h_loc} (batch,head) problems per rank",
        "scaling": "strong",
        "roofline": {"bound": "mfma", "achieved": ach, "peak": PEAK, "unit": "TFLOP/s", "frac": ach / PEAK,
                     "kernel_ms": ms_kernel, "kernel": "attn_fwd_c4_kernel<128,0>",
                     "algorithmic_flops_per_launch": flops_local,
                     "algorithmic_bytes_per_launch": 4.0 * b_loc * h_loc * N * D * 2,
                     "traffic": pmc_traffic("attn_fwd_c4_kernel<128,0>")},
    }


# ---------------------------------------------------------------------------------------------------
def cpu_baseline_hgemm(budget_s: float = 12.0):
    """torch.matmul on fp16 CPU tensors = the reference's `--torch` baseline callable (hgemm.py:1088).
    Bounded sample: a 512^3 probe sets the rate, then the largest cube of the 8192^3 problem whose
    estimated time fits `budget_s` is timed (fp16 CPU matmul speed varies by orders of magnitude
    between hosts)."""
    torch.manual_seed(0)
    a = torch.randn(512, 512, dtype=torch.half)
    torch.matmul(a, a)
    t0 = time.perf_counter()
    torch.matmul(a, a)
    probe = max(time.perf_counter() - t0, 1e-6)
    n, dt = 512, probe
    for cand in (4096, 2048, 1024):     # x4 safety: larger cubes fall out of cache and run slower per FLOP
        if 4.0 * probe * (cand / 512) ** 3 <= budget_s:
            n = cand
            a = torch.randn(n, n, dtype=torch.half)
            b = torch.randn(n, n, dtype=torch.half)
            t0 = time.perf_counter()
            torch.matmul(a, b)
            dt = time.perf_counter() - t0
            break
    out = {"value": 2.0 * n ** 3 / dt * 1e-12, "unit": "TFLOP/s", "cores": torch.get_num_threads(),
           "host_cpus": os.cpu_count(), "kind": "reference",
           "sample": f"torch.matmul fp16 on CPU tensors, M=N=K={n} ({dt:.2f} s; 1/{(8192 // n) ** 3} of the "
                     f"8192^3 work), the reference bench's own torch baseline callable (hgemm.py:1088)"}
    try:  # the C oracle ("port"), fp64 accumulate: 128 output rows of the 8192^3 problem
        from tests import oracle_lib
        orc = oracle_lib.load()
        m, nn = 128, 8192
        a2 = torch.randn(m, nn, dtype=torch.half)
        b2 = torch.randn(nn, nn, dtype=torch.half)
        t0 = time.perf_counter()
        orc.hgemm(a2, b2, m, nn, nn, 0, "exact")
        dt2 = time.perf_counter() - t0
        out["port"] = {"value": 2.0 * m * nn * nn / dt2 * 1e-12, "unit": "TFLOP/s",
                       "cores": orc.lib.lc_oracle_num_threads(), "kind": "port",
                       "sample": f"oracle lc_oracle_hgemm_exact (fp64 accumulate), {m} rows of 8192^3 ({dt2:.2f} s)"}
    except Exception as e:  # the baseline is informational; never fail the bench on it
        out["port"] = {"error": repr(e)}
    return out


def cpu_baseline_attn():
    import torch.nn.functional as F
    B, H, N, D = 1, 4, 4096, 128
    torch.manual_seed(0)
    q, k, v = (torch.randn(B, H, N, D, dtype=torch.half) for _ in range(3))
    F.scaled_dot_product_attention(q[:, :1, :512], k[:, :1, :512], v[:, :1, :512])
    t0 = time.perf_counter()
    F.scaled_dot_product_attention(q, k, v)
    dt = time.perf_counter() - t0
    return {"value": host.mha_matmul_flops(B, H, N, D) / dt * 1e-12, "unit": "TFLOP/s",
            "cores": torch.get_num_threads(), "kind": "reference",
            "sample": f"F.scaled_dot_product_attention fp16 on CPU tensors, B={B} H={H} S={N} D={D} "
                      f"({dt:.2f} s; 1/32 of config 3), the reference bench's sdpa baseline callable"}


# ---------------------------------------------------------------------------------------------------
def main():
    args = parse()
    # stdout carries exactly ONE line (the JSON): park the real stdout and point fd 1 at stderr while
    # libraries (c10d/gloo/RCCL banners, hipBLASLt) may print, restore it for the final print.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    w = lcd.init()
    if w.size != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={w.size}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU path for the HIP kernels")
    capi.load()
    capi.device_check()

    if args.workload == "hgemm":
        main_res = bench_hgemm(w, args)
        extra = None
        if not args.no_attention:
            extra = bench_attn(w, args, sharded_cfg4=False, steps=max(5, args.steps // 5), warmup=1)
    elif args.workload == "attn":
        main_res, extra = bench_attn(w, args, sharded_cfg4=False), None
    else:
        main_res, extra = bench_attn(w, args, sharded_cfg4=True), None

    out = {
        "metric": "achieved fp16 TFLOPS vs MI355X MFMA peak: HGEMM 8192^3; FA-2 fwd S=4096 D=128",
        "value": main_res["value"],
        "unit": "TFLOP/s",
        "n_gpus": w.size,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": main_res["ms_per_step"],
        "higher_is_better": True,
        "scaling": main_res["scaling"],
        "vs_baseline": None,          # BASELINE.md holds no published number for this metric on MI355X
        "dtype": "f16 (fp32 MFMA accumulate)",
        "data": "synthetic",
        "config": {"workload": main_res["workload"], "parallelism": f"{w.size} independent rank(s), no data-path collective"},
        "frac_of_peak": main_res["value"] / (PEAK * w.size),
        "roofline": main_res["roofline"],
    }
    if "vendor_tflops" in main_res:
        out["vendor_tflops"] = main_res["vendor_tflops"]
    if "fp8_tflops" in main_res:
        out["fp8_tflops"] = main_res["fp8_tflops"]
    if extra is not None:
        out["attention"] = {k: extra[k] for k in ("value", "ms_per_step", "tflops_reference_formula", "workload",
                                                   "scaling", "roofline")}
        out["attention"]["frac_of_peak"] = extra["value"] / (PEAK * w.size)
    if w.rank == 0 and w.size == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_hgemm() if args.workload == "hgemm" else cpu_baseline_attn()
        if extra is not None:
            out["attention"]["cpu_baseline"] = cpu_baseline_attn()
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    os.close(real_stdout)
    if w.rank == 0:
        print(json.dumps(out), flush=True)
    os.dup2(2, 1)
    lcd.shutdown(w)


if __name__ == "__main__":
    main()
