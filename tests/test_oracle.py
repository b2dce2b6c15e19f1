"""CPU tests: the oracle against the fixtures generated from the reference's own Python (tests/golden,
tools/make_golden.py) and against independent torch fp64 math; fp16 conversion known answers."""
import math

import numpy as np
import pytest
import torch

from tests import tol


def test_half_conversion_known_answers(oracle):
    lib = oracle.lib
    ka = {0x0000: 0.0, 0x3c00: 1.0, 0xc000: -2.0, 0x7bff: 65504.0, 0x0001: 2.0 ** -24, 0x0400: 2.0 ** -14,
          0x3555: 0.333251953125, 0x03ff: 1023 * 2.0 ** -24}
    for h, f in ka.items():
        assert lib.lc_h2f(h) == f
        assert lib.lc_f2h(f) == h
    assert lib.lc_f2h(float("inf")) == 0x7c00 and lib.lc_f2h(-float("inf")) == 0xfc00
    assert lib.lc_f2h(65520.0) == 0x7c00          # rounds to inf
    assert lib.lc_f2h(65519.0) == 0x7bff
    assert lib.lc_f2h(1.0 + 2.0 ** -11) == 0x3c00  # tie -> even
    assert lib.lc_f2h(1.0 + 3 * 2.0 ** -11) == 0x3c02
    assert lib.lc_d2h(2.0 ** -25) == 0x0000 and lib.lc_d2h(2.0 ** -25 * 1.0000001) == 0x0001


def test_half_conversion_exhaustive_roundtrip(oracle):
    allh = np.arange(65536, dtype=np.uint16)
    f = allh.view(np.float16).astype(np.float32)
    lib = oracle.lib
    for h in range(0, 65536, 7):
        x = lib.lc_h2f(h)
        if math.isnan(float(f[h])):
            assert math.isnan(x)
        else:
            assert x == float(f[h])
            assert lib.lc_f2h(x) == h
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.standard_normal(4000) * 10.0 ** rng.integers(-8, 5, 4000), [65504.0, 1e-8]])
    want = xs.astype(np.float32).astype(np.float16).view(np.uint16)
    got = np.array([lib.lc_f2h(float(np.float32(x))) for x in xs], dtype=np.uint16)
    assert (got == want).all()


@pytest.mark.parametrize("i", range(4))
def test_hgemm_oracle_vs_reference_torch_baseline(oracle, golden, i):
    g = golden["hgemm"]
    a, b, bcol = g[f"a{i}"], g[f"b{i}"], g[f"bcol{i}"]
    M, K = a.shape
    N = b.shape[1]
    c64 = g[f"c64_{i}"]
    for layout, bb in ((0, b), (1, bcol)):
        f32 = oracle.hgemm(a, bb, M, N, K, layout, "f32")
        np.testing.assert_allclose(f32, c64, rtol=1e-6, atol=1e-5)     # oracle == fp64 reference math
        h = oracle.hgemm(a, bb, M, N, K, layout, "exact")
        want = torch.from_numpy(c64.astype(np.float64)).to(torch.half).numpy()
        assert (h.view(np.uint16) != want.view(np.uint16)).mean() < 2e-3  # double rounding via stored fp32
        # the reference's own CPU baseline (torch.matmul fp16, hgemm.py:1088) sits inside the tolerance band
        c16 = g[f"c16_{i}"].view(np.float16).astype(np.float32)
        ok, mx, _ = tol.hgemm_close(c16, f32, K)
        assert ok, mx


def test_hgemm_refnum_is_farther_from_truth_than_fp32_accumulate(oracle):
    rng = np.random.default_rng(3)
    M = N = 64
    K = 2048
    a = rng.standard_normal((M, K)).astype(np.float16)
    b = rng.standard_normal((K, N)).astype(np.float16)
    truth = oracle.hgemm(a, b, M, N, K, 0, "f32")
    ref = oracle.hgemm(a, b, M, N, K, 0, "refnum").astype(np.float32)
    ex = oracle.hgemm(a, b, M, N, K, 0, "exact").astype(np.float32)
    e_ref, e_ex = np.abs(ref - truth).max(), np.abs(ex - truth).max()
    assert e_ex <= 0.5 * 2.0 ** -10 * np.abs(truth).max() + 1e-6   # one fp16 rounding
    assert e_ref > 4 * e_ex                                        # fp16 accumulation is visibly worse
    assert e_ref < 2.0                                             # but not garbage: the emulator is sane


def test_oracle_tn_equals_nn(oracle):
    rng = np.random.default_rng(4)
    M, N, K = 48, 80, 72
    a = rng.standard_normal((M, K)).astype(np.float16)
    b = rng.standard_normal((K, N)).astype(np.float16)
    bt = np.ascontiguousarray(b.T)
    x = oracle.hgemm(a, b, M, N, K, 0, "exact")
    y = oracle.hgemm(a, bt, M, N, K, 1, "exact")
    assert (x.view(np.uint16) == y.view(np.uint16)).all()


@pytest.mark.parametrize("i", range(4))
def test_attn_oracle_vs_reference_unfused(oracle, golden, i):
    g = golden["attn"]
    q, k, v = g[f"q{i}"], g[f"k{i}"], g[f"v{i}"]
    B, H, N, D = q.shape
    o = oracle.attn(q, k, v, B, H, N, D, mode="f32")
    np.testing.assert_allclose(o, g[f"o64_{i}"], rtol=2e-6, atol=2e-6)  # == unfused_standard_attn in fp64
    vt = np.ascontiguousarray(np.swapaxes(v.view(np.float16), -1, -2))
    ot = oracle.attn(q, k, vt, B, H, N, D, vt=True, mode="f32")
    np.testing.assert_allclose(ot, o, rtol=1e-6, atol=1e-7)
    # the reference's fp16 CPU baselines agree with the oracle inside its own --check threshold
    for nm in ("o16", "sdpa16"):
        ref16 = g[f"{nm}{i}"].view(np.float16).astype(np.float32)
        assert np.abs(ref16 - o).max() < tol.ATTN_ATOL


def test_attn_refnum_envelope(oracle):
    """The restated split-Q numerics (fp16 accumulate, fp16 P, fp16/fp32 running O) stay inside the
    reference's --check threshold and are no closer to truth than one fp16 rounding."""
    rng = np.random.default_rng(5)
    B, H, N, D = 1, 2, 256, 64
    q, k, v = (rng.standard_normal((B, H, N, D)).astype(np.float16) for _ in range(3))
    truth = oracle.attn(q, k, v, B, H, N, D, mode="f32")
    for bc, of32 in ((16, False), (64, True)):
        r = oracle.attn(q, k, v, B, H, N, D, mode="refnum", Bc=bc, o_f32=of32).astype(np.float32)
        e = np.abs(r - truth).max()
        assert e < tol.ATTN_ATOL
        assert np.isfinite(r).all()


def test_attn_rows_subset_matches_full(oracle):
    rng = np.random.default_rng(6)
    BH, N, D = 3, 128, 32
    q, k, v = (rng.standard_normal((BH, N, D)).astype(np.float16) for _ in range(3))
    full = oracle.attn(q, k, v, BH, 1, N, D, mode="f32").reshape(BH, N, D)
    rows = [5, 17, 100]
    sub = oracle.attn_rows(np.ascontiguousarray(q[:, rows]), k, v, BH, len(rows), N, D)
    np.testing.assert_array_equal(sub, full[:, rows])


def test_flop_accounting_matches_reference(oracle, golden):
    for B, H, N, D, secs, om, want in golden["host"]["mha_tflops"]:
        got = oracle.lib.lc_oracle_mha_flops(B, H, N, D, int(om)) * 1e-12 / secs
        assert got == pytest.approx(want, rel=1e-12)
    assert oracle.lib.lc_oracle_hgemm_flops(8192, 8192, 8192) == 2.0 * 8192 ** 3
    for N, K, f, want in golden["host"]["swizzle_stride"]:
        assert oracle.lib.lc_oracle_block_swizzle_stride(N, K, -1.0 if f is None else f) == want


def test_e4m3_decoder_matches_torch_float8(oracle):
    """OCP e4m3fn decode pinned on torch's own float8_e4m3fn -> float32 conversion, all 256 codes."""
    codes = torch.arange(256, dtype=torch.uint8)
    want = codes.view(torch.float8_e4m3fn).float().numpy()
    for c in range(256):
        got = oracle.lib.lc_e4m3_to_f32(c)
        if np.isnan(want[c]):
            assert np.isnan(got)
        else:
            assert got == want[c], (c, got, want[c])


def test_fp8_gemm_oracle_vs_torch(oracle):
    torch.manual_seed(12)
    M, N, K = 48, 40, 136
    a = (torch.randn(M, K) * 0.5).to(torch.float8_e4m3fn)
    b = (torch.randn(N, K) * 0.5).to(torch.float8_e4m3fn)
    got = oracle.gemm_fp8(a, b, M, N, K, alpha=0.25)
    want = 0.25 * (a.double() @ b.double().t())
    np.testing.assert_allclose(got, want.float().numpy(), rtol=1e-6, atol=1e-6)


def test_mxfp8_gemm_oracle_vs_torch(oracle):
    """The MX restatement (E8M0 block scale per row and 32 k) pinned on torch fp64 arithmetic over the dequantised operands; unit
    scales (127) reduce it to the plain fp8 oracle."""
    torch.manual_seed(14)
    M, N, K = 40, 24, 160
    a = (torch.randn(M, K) * 0.5).to(torch.float8_e4m3fn)
    b = (torch.randn(N, K) * 0.5).to(torch.float8_e4m3fn)
    sa = torch.randint(120, 135, (M, K // 32), dtype=torch.uint8)
    sb = torch.randint(120, 135, (N, K // 32), dtype=torch.uint8)
    got = oracle.gemm_mxfp8(a, sa, b, sb, M, N, K, alpha=0.5)
    da = a.double() * torch.pow(2.0, sa.double() - 127).repeat_interleave(32, dim=1)
    db = b.double() * torch.pow(2.0, sb.double() - 127).repeat_interleave(32, dim=1)
    np.testing.assert_allclose(got, (0.5 * (da @ db.t())).float().numpy(), rtol=1e-6, atol=1e-6)
    one = torch.full_like(sa, 127), torch.full_like(sb, 127)
    np.testing.assert_array_equal(oracle.gemm_mxfp8(a, one[0], b, one[1], M, N, K, alpha=0.5), oracle.gemm_fp8(a, b, M, N, K, alpha=0.5))


def test_bf16_attention_oracle_vs_torch(oracle):
    torch.manual_seed(13)
    B, H, N, D = 1, 2, 64, 256
    q, k, v = (torch.randn(B, H, N, D).to(torch.bfloat16) for _ in range(3))
    got = oracle.attn_bf16(q, k, v, B, H, N, D)
    att = torch.softmax(q.double() @ k.double().transpose(-2, -1) / (D ** 0.5), dim=-1) @ v.double()
    np.testing.assert_allclose(got, att.float().numpy(), rtol=2e-6, atol=2e-6)
