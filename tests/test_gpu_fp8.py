"""GPU parity of the fp8 (e4m3fn) GEMM extension (BASELINE config 5) against the fp64 oracle."""
import numpy as np
import pytest
import torch

from tests import tol

pytestmark = pytest.mark.gpu


def _capi():
    from leetcuda_amd import capi
    capi.load()
    return capi


@pytest.fixture(params=[1, 2, 0], ids=["mx_k64_4wave", "mx_k64_8wave", "plain_k16"])
def mx(request):
    """Both MFMA forms of the fp8 kernel: v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales (default) and
    v_mfma_f32_32x32x16_fp8_fp8 (lc_tune_set "fp8_mx")."""
    capi = _capi()
    capi.tune("fp8_mx", request.param)
    yield request.param
    capi.tune("fp8_mx", 1)


@pytest.mark.parametrize("shape", [(256, 256, 128), (512, 256, 384), (256, 768, 1024), (1024, 1024, 2048)])
def test_fp8_gemm_vs_oracle(oracle, shape, mx):
    capi = _capi()
    M, N, K = shape
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda").to(torch.float8_e4m3fn)
    b = torch.randn(N, K, device="cuda").to(torch.float8_e4m3fn)        # stored [N,K] (TN)
    for alpha, stride in ((1.0, 1), (0.125, 512)):
        c = torch.full((M, N), float("nan"), dtype=torch.half, device="cuda")
        capi.gemm_fp8(a, b, c, alpha=alpha, swizzle_stride=stride)
        torch.cuda.synchronize()
        truth = oracle.gemm_fp8(a, b, M, N, K, alpha)
        ok, mx, ex = tol.hgemm_close(c.float().cpu().numpy(), truth, K, atol=tol.fp8_atol(K, alpha ** 0.5))
        assert ok, (mx, ex)


def test_fp8_identity_detects_transposes(mx):
    capi = _capi()
    n = 512
    eye = torch.eye(n, device="cuda").to(torch.float8_e4m3fn)
    vals = torch.tensor([0.5, 1.0, 1.5, 2.0, 3.0, -4.0, 6.0, -0.25], device="cuda")
    b = vals[(torch.arange(n * n, device="cuda").reshape(n, n) * 7 // 3) % 8].to(torch.float8_e4m3fn)  # asymmetric
    c = torch.zeros(n, n, dtype=torch.half, device="cuda")
    capi.gemm_fp8(eye, b, c)            # C = I * B^T  (b stored [N,K])
    torch.cuda.synchronize()
    assert torch.equal(c, b.float().t().half())
    capi.gemm_fp8(b, eye, c)            # C = B * I
    torch.cuda.synchronize()
    assert torch.equal(c, b.float().half())


def test_fp8_config5_16384_properties(oracle, mx):
    """BASELINE config 5: M=N=K=16384 fp8: sampled rows vs the exact oracle + C*x == A*(B^T*x)."""
    capi = _capi()
    n = 16384
    torch.manual_seed(0)
    a = torch.randn(n, n, device="cuda").to(torch.float8_e4m3fn)
    b = torch.randn(n, n, device="cuda").to(torch.float8_e4m3fn)
    c = torch.zeros(n, n, dtype=torch.half, device="cuda")
    alpha = 1.0 / 16
    capi.gemm_fp8(a, b, c, alpha=alpha, swizzle_stride=4096)
    torch.cuda.synchronize()
    rows = [0, 255, 256, 8191, 16383]
    truth = oracle.gemm_fp8(a[rows].contiguous(), b, len(rows), n, n, alpha)
    ok, mx, _ = tol.hgemm_close(c[rows].float().cpu().numpy(), truth, n, atol=tol.fp8_atol(n, alpha ** 0.5))
    assert ok, mx
    x = torch.randn(n, device="cuda", dtype=torch.float64)
    want = alpha * (a.double() @ (b.double().t() @ x))          # checker math on the GPU in fp64 (torch, not ours)
    got = c.double() @ x
    rel = ((got - want).abs().max() / want.abs().max()).item()
    assert rel < 2e-3, rel
