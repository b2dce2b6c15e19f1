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


@pytest.fixture(params=[3, 1, 2, 0], ids=["mx_k128_generated", "mx_k64_4wave", "mx_k64_8wave", "plain_k16"])
def mx(request):
    """Every MFMA form of the fp8 kernel: v_mfma_scale_f32_16x16x128_f8f6f4 in the generated loop (default), v_mfma_scale_f32_32x32x64_f8f6f4
    with unit block scales and v_mfma_f32_32x32x16_fp8_fp8 (lc_tune_set "fp8_mx")."""
    capi = _capi()
    capi.tune("fp8_mx", request.param)
    yield request.param
    capi.tune("fp8_mx", 3)


@pytest.mark.parametrize("shape", [(256, 256, 128), (512, 256, 384), (256, 768, 1024), (1024, 1024, 2048), (256, 512, 256), (4096, 8192, 640)])
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


# ---- OCP MX block scales (lc_gemm_mxfp8) ------------------------------------------------------------------------------------------

def _mx_inputs(M, N, K, seed, spread):
    """randn e4m3 data and E8M0 scales drawn from 127 - spread .. 127 + spread per (row, 32 k)."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.float8_e4m3fn)
    b = torch.randn(N, K, device="cuda", generator=g).to(torch.float8_e4m3fn)
    sa = torch.randint(127 - spread, 128 + spread, (M, K // 32), device="cuda", generator=g, dtype=torch.uint8)
    sb = torch.randint(127 - spread, 128 + spread, (N, K // 32), device="cuda", generator=g, dtype=torch.uint8)
    return a, sa, b, sb


MX_ALIGN = 2.0 ** -12   # see _mx_bound


def _mx_bound(a, sa, b, sb, rows, alpha):
    """|err| allowance per output for the matrix core's block sum: one v_mfma_scale_f32_16x16x128 aligns its 128 products to the largest
    and keeps a fixed number of bits below it (tools/cpp/mx_probe.cpp: exact on narrow-range data, up to 1e-4 of sum |terms| on data
    spanning 2^28), so the allowance is MX_ALIGN * (sum over K tiles of the largest |a_k b_k| of that tile) — computed in fp64 with torch on
    the dequantised operands (checker math).  Returns a numpy array [len(rows), N]."""
    da = (a[rows].double() * torch.pow(2.0, sa[rows].double() - 127).repeat_interleave(32, dim=1)).abs()
    db = (b.double() * torch.pow(2.0, sb.double() - 127).repeat_interleave(32, dim=1)).abs()
    out = torch.zeros(len(rows), b.shape[0], dtype=torch.float64, device=a.device)
    for k0 in range(0, a.shape[1], 128):
        for r0 in range(0, len(rows), 8):   # [8, N, 128] fp64 at a time
            out[r0:r0 + 8] += (da[r0:r0 + 8, None, k0:k0 + 128] * db[None, :, k0:k0 + 128]).amax(dim=2)
    return (abs(alpha) * MX_ALIGN * out).cpu().numpy()


@pytest.mark.parametrize("shape", [(256, 256, 128), (256, 512, 384), (512, 256, 1024), (1024, 1024, 2048), (4096, 8192, 640)])
@pytest.mark.parametrize("spread", [0, 3, 12])
def test_mxfp8_gemm_vs_oracle(oracle, shape, spread):
    """lc_gemm_mxfp8 against the fp64 MX oracle with NON-UNIT block scales (spread 3: 2^-3 .. 2^3 per block and operand, 12: 2^-12 .. 2^12);
    spread 0 (all scales 1.0) must reproduce lc_gemm_fp8_e4m3's K = 128 kernel bit for bit."""
    capi = _capi()
    M, N, K = shape
    a, sa, b, sb = _mx_inputs(M, N, K, M + N + K + spread, spread)
    pa, pb = capi.mxfp8_pack_scales(sa), capi.mxfp8_pack_scales(sb)
    alpha = 2.0 ** -(2 + spread)
    c = torch.full((M, N), float("nan"), dtype=torch.half, device="cuda")
    capi.gemm_mxfp8(a, pa, b, pb, c, alpha=alpha, swizzle_stride=512)
    torch.cuda.synchronize()
    rows = list(range(0, M, 37))[:32] + [M - 1]
    truth = oracle.gemm_mxfp8(a[rows].contiguous(), sa[rows].contiguous(), b, sb, len(rows), N, K, alpha)
    err = np.abs(c[rows].float().cpu().numpy().astype(np.float64) - truth)
    bound = 2.0 ** -11 * np.abs(truth) + _mx_bound(a, sa, b, sb, rows, alpha) + 1e-7
    assert (err <= bound).all(), (float(err.max()), float((err - bound).max()))
    if spread == 0:
        c1 = torch.empty_like(c)
        capi.gemm_fp8(a, b, c1, alpha=alpha, swizzle_stride=512)
        torch.cuda.synchronize()
        assert torch.equal(c, c1)


def test_mxfp8_scales_reach_their_block():
    """One block scale at a time: doubling the scale of (row r, k block kb) of A must change row r of C by exactly the contribution of that
    block — checks the packed layout lane by lane (every fragment, register half and K tile position) with exactly representable data."""
    capi = _capi()
    M, N, K = 256, 256, 512
    g = torch.Generator(device="cuda").manual_seed(5)
    vals = torch.tensor([0.5, 1.0, 1.5, -1.0, 2.0, -0.5, 0.75, -2.0], device="cuda")
    a = vals[torch.randint(0, 8, (M, K), device="cuda", generator=g)].to(torch.float8_e4m3fn)
    b = vals[torch.randint(0, 8, (N, K), device="cuda", generator=g)].to(torch.float8_e4m3fn)
    sb = torch.full((N, K // 32), 127, dtype=torch.uint8, device="cuda")
    pb = capi.mxfp8_pack_scales(sb)
    base = a.float() @ b.float().t()
    c = torch.empty(M, N, dtype=torch.half, device="cuda")
    cases = [(r, kb) for r in (0, 1, 15, 16, 63, 64, 100, 127, 128, 200, 255) for kb in (0, 1, 2, 3, 4, 7, 10, 15)]
    for r, kb in cases:
        sa = torch.full((M, K // 32), 127, dtype=torch.uint8, device="cuda")
        sa[r, kb] = 128
        capi.gemm_mxfp8(a, capi.mxfp8_pack_scales(sa), b, pb, c, alpha=1 / 16)
        want = base.clone()
        want[r] += a[r, 32 * kb:32 * kb + 32].float() @ b[:, 32 * kb:32 * kb + 32].float().t()
        torch.cuda.synchronize()
        assert torch.equal(c, (want / 16).half()), (r, kb)
    # and the B side (scale_src0 of the instruction: the operands are swapped inside the kernel)
    sa = torch.full((M, K // 32), 127, dtype=torch.uint8, device="cuda")
    pa = capi.mxfp8_pack_scales(sa)
    for r, kb in cases[::5]:
        sb2 = sb.clone()
        sb2[r, kb] = 126
        capi.gemm_mxfp8(a, pa, b, capi.mxfp8_pack_scales(sb2), c, alpha=1 / 16)
        want = base.clone()
        want[:, r] -= 0.5 * (a[:, 32 * kb:32 * kb + 32].float() @ b[r, 32 * kb:32 * kb + 32].float())
        torch.cuda.synchronize()
        assert torch.equal(c, (want / 16).half()), (r, kb)


def test_mxfp8_config5_16384_properties(oracle):
    """BASELINE config 5's size with real block scales: sampled rows vs the MX oracle + the linearity check C x == A' (B'^T x)."""
    capi = _capi()
    n = 16384
    a, sa, b, sb = _mx_inputs(n, n, n, 0, 2)
    pa, pb = capi.mxfp8_pack_scales(sa), capi.mxfp8_pack_scales(sb)
    c = torch.zeros(n, n, dtype=torch.half, device="cuda")
    alpha = 1.0 / 64
    capi.gemm_mxfp8(a, pa, b, pb, c, alpha=alpha, swizzle_stride=4096)
    torch.cuda.synchronize()
    rows = [0, 255, 256, 8191, 16383]
    truth = oracle.gemm_mxfp8(a[rows].contiguous(), sa[rows].contiguous(), b, sb, len(rows), n, n, alpha)
    err = np.abs(c[rows].float().cpu().numpy().astype(np.float64) - truth)
    bound = 2.0 ** -11 * np.abs(truth) + _mx_bound(a, sa, b, sb, rows, alpha) + 1e-7
    assert (err <= bound).all(), (float(err.max()), float((err - bound).max()))
    x = torch.randn(n, device="cuda", dtype=torch.float64)
    w = torch.zeros(n, device="cuda", dtype=torch.float64)
    for r0 in range(0, n, 2048):   # B'^T x in row blocks (fp64 dequantised B would be 2 GiB at once: fine, but keep the peak low)
        db = b[r0:r0 + 2048].double() * torch.pow(2.0, sb[r0:r0 + 2048].double() - 127).repeat_interleave(32, dim=1)
        w += db.t() @ x[r0:r0 + 2048]
    want = torch.empty(n, device="cuda", dtype=torch.float64)
    for r0 in range(0, n, 2048):
        da = a[r0:r0 + 2048].double() * torch.pow(2.0, sa[r0:r0 + 2048].double() - 127).repeat_interleave(32, dim=1)
        want[r0:r0 + 2048] = alpha * (da @ w)
    got = c.double() @ x
    rel = ((got - want).abs().max() / want.abs().max()).item()
    assert rel < 2e-3, rel
