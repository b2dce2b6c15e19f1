"""Tolerances of the parity tests, stated once.

north_star: "outputs that match the reference CUDA kernels' results on identical random inputs within
1e-2 rtol for fp16".  The reference kernels accumulate in fp16 and cannot run here, so parity is judged
against the exact result of the reference's definition (oracle *_exact):

  HGEMM      |out - truth| <= HGEMM_RTOL*|truth| + hgemm_atol(K, amp)
             rtol 1e-2 is the north_star figure; the absolute term covers fp32 accumulation noise at
             truth ~ 0 where a relative bound is ill-posed (K/32 sequential fp32 MFMA accumulations of
             partial sums ~ amp^2*sqrt(K): <= ~1e-3 at K = 8192, randn inputs).
  attention  torch.allclose(atol=1e-2) is the reference's own check (flash_attn_mma.py:489); we also hold
             the README's error envelope "max < ~1e-3" (README.md:130) with ATTN_MAX_ABS.
"""
import math

HGEMM_RTOL = 1e-2
ATTN_ATOL = 1e-2      # the reference's --check threshold
ATTN_MAX_ABS = 2e-3   # tighter: what fp32-accumulate kernels should hold on randn inputs AT N <= 256
ATTN_RTOL_F16 = 2.0 ** -10   # one fp16 ulp of |truth|: the final rounding of O (half an ulp) + headroom
ATTN_RTOL_BF16 = 2.0 ** -7   # the same for bfloat16 outputs


def attn_max_abs(N: int, bf16: bool = False) -> float:
    """Absolute part of the long-sequence attention bound, scaled with the signal (round-3 verdict, weak #1): on randn
    inputs |O| ~ 1/sqrt(N) and the kernel's error (fp16 rounding of P averaged over N keys) shrinks the same way — smoke
    measures 2.4e-4 at N = 256, ~4e-5 at N = 8192 — so a fixed 2e-3 at N = 8192 would let a 25 % mis-scaling of one 64-key
    tile (dO <= 1.5e-3) through.  ATTN_MAX_ABS * sqrt(256 / N) (3.5e-4 at N = 8192) does not; the bound a test applies is
    |out - truth| <= attn_max_abs(N) + ATTN_RTOL * |truth| (the relative term is the output's own fp16 / bf16 rounding,
    which matters only on rows a spike input drives to |O| >> 1/sqrt(N))."""
    base = 1.6e-2 if bf16 else ATTN_MAX_ABS
    return base * math.sqrt(256.0 / max(N, 256))


ATTN_RTOL_SPIKE = 2.0 ** -8   # heads carrying a planted spike key (|score| ~ 3 sqrt(D) >> 1): Q~ = fp16(Q * scale * log2e) and K are fp16,
                              # so a score carries a relative error ~2^-11 and a weight e^s a relative error ~|s| 2^-11 — measured 2.2e-3
                              # at |O| = 1.5 on the D = 32 spike head (r4a); the reference's fp16-accumulated Q.K^T is coarser still


def attn_close(out_f32, truth_f32, N: int, bf16: bool = False, rtol=None):
    """numpy arrays -> (ok, max_abs_err, worst_excess) under the N-scaled bound."""
    import numpy as np
    err = np.abs(out_f32.astype(np.float64) - truth_f32.astype(np.float64))
    if rtol is None:
        rtol = ATTN_RTOL_BF16 if bf16 else ATTN_RTOL_F16
    bound = attn_max_abs(N, bf16) + rtol * np.abs(truth_f32.astype(np.float64))
    return bool((err <= bound).all()), float(err.max()), float((err - bound).max())


def hgemm_atol(K: int, amp: float = 1.0) -> float:
    return amp * amp * (1e-3 + 2.5e-7 * K)


def fp8_atol(K: int, amp: float = 1.0) -> float:
    """fp8 MFMA sums its 16 products per instruction with a narrower internal alignment than a chain of fp32
    FMAs: measured |err| at truth ~ 0 is ~1.7e-3 at K = 2048 (randn e4m3 inputs), ~10x the fp16 kernels'."""
    return amp * amp * (2e-3 + 1.5e-6 * K)


def hgemm_close(out_f32, truth_f32, K, amp=1.0, atol=None):
    """numpy arrays -> (ok, max_abs_err, worst_excess)."""
    import numpy as np
    err = np.abs(out_f32.astype(np.float64) - truth_f32.astype(np.float64))
    bound = HGEMM_RTOL * np.abs(truth_f32.astype(np.float64)) + (hgemm_atol(K, amp) if atol is None else atol)
    return bool((err <= bound).all()), float(err.max()), float((err - bound).max())
