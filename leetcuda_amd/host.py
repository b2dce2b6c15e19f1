"""Host-side helpers mirrored from the reference's bench drivers (same names, argument meaning and
results) so parity tests and benches read like the reference's own scripts.

  as_col_major               kernels/hgemm/tools/utils.py:152-156
  make_block_swizzle_stride  kernels/hgemm/hgemm.py:198-208
  hgemm_tflops               kernels/hgemm/hgemm.py:282
  get_mha_tflops             kernels/flash-attn/flash_attn_mma.py:241-278
  get_qkvo                   kernels/flash-attn/flash_attn_mma.py:417-445  (layouts only)
  shard_bounds               batch / batch*heads sharding of the attention path
"""
from __future__ import annotations

import torch

MI355X_FP16_DENSE_PEAK_TFLOPS = 2500.0  # /opt/skills/guides/MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
MI355X_FP8_MX_DENSE_PEAK_TFLOPS = 5000.0  # same table: "Peak FP8 MFMA ~5 PF dense" (MX-scaled K = 64 / 128 forms)
MI355X_HBM_PEAK_GBS = 8000.0


@torch.no_grad()
def as_col_major(x: torch.Tensor) -> torch.Tensor:
    """Row-major [K,N] -> same shape whose storage is the column-major (i.e. [N,K] row-major) image."""
    return x.t().reshape(x.shape).contiguous()


def make_block_swizzle_stride(N: int, K: int, swizzle_factor: float | None = None) -> int:
    if swizzle_factor is None:
        swizzle_factor = 0.5 if N <= 4096 else 0.25
        if all((N >= 14848, K > 8192, N % 8 == 0)):
            swizzle_factor = 0.125
    swizzle_stride = int(N * swizzle_factor)
    return swizzle_stride if swizzle_stride >= 256 else 1


def hgemm_tflops(M: int, N: int, K: int, secs: float) -> float:
    return (2 * M * N * K) * 1e-12 / secs


def get_mha_tflops(B: int, H: int, N: int, D: int, secs: float = 1.0, only_matmul: bool = False) -> float:
    flops_qk = B * H * N * N * (2 * D - 1)
    flops_scaling = B * H * N * N
    flops_row_max = B * H * N * (N - 1)
    flops_subtract_max = B * H * N * N
    flops_exp = B * H * N * N
    flops_row_sum = B * H * N * (N - 1)
    flops_normalization = B * H * N * N
    flops_safe_softmax = flops_row_max + flops_subtract_max + flops_exp + flops_row_sum + flops_normalization
    flops_pv = B * H * N * D * (2 * N - 1)
    total = flops_qk + flops_scaling + flops_safe_softmax + flops_pv
    if only_matmul:
        total = flops_qk + flops_pv
    return total * 1e-12 / secs


def mha_matmul_flops(B: int, H: int, N: int, D: int) -> float:
    """4*B*H*N^2*D — the roofline numerator (MFMA work only)."""
    return 4.0 * B * H * N * N * D


def get_qkvo(B, H, N, D, device="cuda", seed=None):
    """q,k,v,o [B,H,N,D] fp16 (+ tv = V transposed [B,H,D,N] for the *_swizzle_qkv entries)."""
    g = None
    if seed is not None:
        g = torch.Generator(device=device)
        g.manual_seed(seed)
    q = torch.randn((B, H, N, D), dtype=torch.half, device=device, generator=g)
    k = torch.randn((B, H, N, D), dtype=torch.half, device=device, generator=g)
    v = torch.randn((B, H, N, D), dtype=torch.half, device=device, generator=g)
    o = torch.zeros(B, H, N, D, device=device, dtype=torch.half)
    tv = v.transpose(-2, -1).contiguous()
    return q, k, v, o, tv


def shard_bounds(total: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous slice [lo, hi) of `total` independent units for `rank` (remainder to the low ranks)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def attn_shard(B: int, H: int, world: int, rank: int) -> tuple[int, int, int]:
    """Shard the attention path: whole batches when world | B, otherwise the flattened B*H axis.
    Returns (b_local, h_local, first_unit) where the rank owns b_local*h_local (batch,head) problems."""
    if B % world == 0:
        lo, hi = shard_bounds(B, world, rank)
        return hi - lo, H, lo * H
    lo, hi = shard_bounds(B * H, world, rank)
    return 1, hi - lo, lo
