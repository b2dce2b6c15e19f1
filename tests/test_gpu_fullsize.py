"""Full-size GPU parity that round 3's verdict found missing (VERDICT r3 "what's weak" 1, 4 and "missing" 5):

  * one KV tile perturbed at S = 8192, per kernel family: K rows of ONE 64-key tile scaled by 1.25 (the first tile, the tile
    where the 4-slot LDS ring wraps, the last tile) — that tile's scores stand out, so a stale max, a skipped rescale or a
    mis-addressed ring slot confined to it moves the sampled rows beyond the N-scaled bound of tests/tol.py;
  * fp16 HGEMM at the sizes the reference publishes (kernels/hgemm/README.md:159-185: 12544 ... 16384), NN and TN, default
    knobs: that is where the XCD super-block raster, the ragged-tail split and the non-persistent launch combine;
  * full-size parity for D = 256 at (1,48,8192,256), a V-transposed entry at config 3, D = 1024 at N >= 2048.

Reference entry points: flash_attn_mma.py:465-494 (the --check recipe), flash_attn_mma_share_qkv_swizzle_qkv.cu:961-1010 (V as
[B,H,D,N]), flash_attn_mma_tiling_qkv.cu:881-945 (head dims up to 1024).
"""
import numpy as np
import pytest
import torch

from leetcuda_amd import host
from tests import tol
from tests.test_gpu_configs import _rows_for, _sampled_rows_check

pytestmark = pytest.mark.gpu


def _capi():
    from leetcuda_amd import capi
    capi.load()
    return capi


# (label, D, knob key, knob value): every attention kernel family that can take an S = 8192 problem
FAMILIES = [
    ("d128-auto", 128, None, 0),
    ("d128-one-block", 128, "attn_nw", 513),
    ("d128-generated", 128, "attn_nw", 514),
    ("d128-persistent-static", 128, "attn_nw", 515),
    ("d128-persistent-queue", 128, "attn_nw", 517),
    ("d128-lockstep8", 128, "attn_nw", 8),
    ("d64-auto", 64, None, 0),
    ("d64-generated", 64, "attn_nw", 514),
    ("d64-one-block", 64, "attn_nw", 513),
    ("d64-persistent-static", 64, "attn_nw", 515),
    ("d96-auto", 96, None, 0),
    ("d32-auto", 32, None, 0),
    ("d256-auto", 256, None, 0),
    ("d256-other-mfma-shape", 256, "attn_d512", 3),
    ("d256-rings", 256, "attn_d512", 4),          # attn_bigd7 (auto hands this test's small grid — 2 heads — to attn_bigd2)
    ("d512-auto", 512, None, 0),
    ("d512-column-split", 512, "attn_d512", 1),
    ("d512-other-mfma-shape", 512, "attn_d512", 3),
    ("d1024-pair", 1024, None, 0),
    ("d1024-column-split", 1024, "attn_d512", 1),
    ("d128-v-transposed", -128, None, 0),       # D < 0: V handed over as [B,H,D,N] through a *_swizzle_qkv entry
    ("d64-v-transposed", -64, None, 0),
    ("d256-v-transposed", -256, None, 0),
    ("d256-v-transposed-other-mfma-shape", -256, "attn_d512", 3),
    ("d256-v-transposed-rings", -256, "attn_d512", 4),
]


@pytest.mark.parametrize("label,D,key,val", FAMILIES, ids=[f[0] for f in FAMILIES])
def test_one_kv_tile_perturbed_s8192(oracle, label, D, key, val):
    capi = _capi()
    B, H, N = 1, 2, 8192
    T = N // 64
    vt, D = D < 0, abs(D)
    torch.manual_seed(1000 + D)
    q = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    k0 = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    v = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    rows = _rows_for(N, D)
    heads = [(0, 0), (0, 1)]
    entry = "flash_attn_mma_stages_split_q" if D <= 128 else "flash_attn_mma_stages_split_q_tiling_qkv"
    if vt:
        entry = "flash_attn_mma_stages_split_q_shared_qkv_swizzle_qkv" if D <= 128 else "flash_attn_mma_stages_split_q_tiling_qk_swizzle_qkv"
        v = v.transpose(-2, -1).contiguous()
    outs = []
    for t in (0, 4, T - 1):           # first tile, the tile that re-uses ring slot 0, last tile
        k = k0.clone()
        k[:, :, 64 * t:64 * t + 64] *= 1.25
        o = torch.full_like(q, float("nan"))
        if key:
            capi.tune(key, val)
        try:
            capi.attn_call(entry, q, k, v, o, 2)
            torch.cuda.synchronize()
        finally:
            if key:
                capi.tune(key, 0)
        assert torch.isfinite(o).all(), (label, t)
        _sampled_rows_check(oracle, q, k, v, o, heads, rows, vt=vt)
        outs.append(o)
    # the three perturbations are different problems: the outputs must differ (a kernel that ignored K's tile would not)
    assert not torch.equal(outs[0], outs[1]) and not torch.equal(outs[1], outs[2])


@pytest.mark.parametrize("layout", ["nn", "tn"])
@pytest.mark.parametrize("n", [12544, 15616, 16384])
def test_hgemm_reference_published_sizes(oracle, n, layout):
    """kernels/hgemm/README.md:159-185 publishes 12544 ... 16384 (M = N = K).  Default knobs (auto raster = XCD super-block beyond
    the Infinity Cache, tail split when the last wave is short, persistent walk only when the tiles divide evenly), the
    reference's primary entry names."""
    capi = _capi()
    lay = capi.LAYOUT_NN if layout == "nn" else capi.LAYOUT_TN
    torch.manual_seed(n)
    a = torch.randn(n, n, dtype=torch.half, device="cuda")
    b = torch.randn(n, n, dtype=torch.half, device="cuda")
    bb = host.as_col_major(b) if lay == capi.LAYOUT_TN else b
    c = torch.full((n, n), float("nan"), dtype=torch.half, device="cuda")
    name = ("hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem" if layout == "nn"
            else "hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_tn_swizzle_x4")
    capi.hgemm_call(name, a, bb, c, 2, True, host.make_block_swizzle_stride(n, n))
    torch.cuda.synchronize()
    assert capi.hgemm_kernel_name(n, n, n, lay).startswith("hgemm_w4y_kernel")
    assert torch.isfinite(c).all()
    # (1) C.x == A.(B.x) in fp64 on the host (row chunks: the fp64 copies of 512 MiB operands are never materialised whole)
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n)

    def matvec(t, vec):
        out = np.empty(t.shape[0])
        for r0 in range(0, t.shape[0], 2048):
            out[r0:r0 + 2048] = t[r0:r0 + 2048].cpu().numpy().astype(np.float64) @ vec
        return out
    want = matvec(a, matvec(b, x))
    got = matvec(c, x)
    rel = np.abs(got - want).max() / np.abs(want).max()
    assert rel < 2e-3, rel
    # (2) sampled rows (first / last tile rows, a tail-split tile, mid-matrix) against the exact oracle over the full K
    rows = [0, 255, 256, n // 2 + 1, n - 257, n - 1]
    truth = oracle.hgemm(a[rows].contiguous(), b, len(rows), n, n, 0, "f32")
    ok, mx, _ = tol.hgemm_close(c[rows].float().cpu().numpy(), truth, n)
    assert ok, mx
    # (3) the vendor comparator (hipBLASLt, fp32 compute) agrees on the same rows and to fp16 rounding everywhere
    capi.vendor_init()
    try:
        cv = torch.empty_like(c)
        capi.hgemm_vendor(a, bb, cv, lay)
        torch.cuda.synchronize()
        ok, mx, _ = tol.hgemm_close(cv[rows].float().cpu().numpy(), truth, n)
        assert ok, mx
        ulp = torch.clamp(cv.float().abs(), min=64.0) * 2.0 ** -10
        assert ((c.float() - cv.float()).abs() <= ulp).all()
    finally:
        capi.vendor_destroy()


@pytest.mark.parametrize("layout", ["nn", "tn"])
@pytest.mark.parametrize("shape", [(8320, 8320, 8320), (8192, 8320, 8192), (8320, 8192, 8192), (8192, 8192, 8224)])
def test_hgemm_reference_legal_shapes_full_size(oracle, shape, layout):
    """Round-4 verdict (next #2): M, N multiples of 128 (not 256) and K % 64 == 32 at 8192-class sizes, NN + TN, default knobs,
    through the reference's primary entry names — flagship kernel on the 256-tileable interior, 128-wide border strips on the
    128-tile kernel, the K remainder as a half K-step (reference: hgemm_mma_stage.cu:650,675-676)."""
    capi = _capi()
    M, N, K = shape
    lay = capi.LAYOUT_NN if layout == "nn" else capi.LAYOUT_TN
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, dtype=torch.half, device="cuda")
    b = torch.randn(K, N, dtype=torch.half, device="cuda")
    bb = host.as_col_major(b) if lay == capi.LAYOUT_TN else b
    c = torch.full((M, N), float("nan"), dtype=torch.half, device="cuda")
    name = ("hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem" if layout == "nn"
            else "hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_tn_swizzle_x4")
    capi.hgemm_call(name, a, bb, c, 2, True, host.make_block_swizzle_stride(N, K))
    torch.cuda.synchronize()
    assert capi.hgemm_kernel_name(M, N, K, lay).startswith("hgemm_w4y_kernel")
    assert torch.isfinite(c).all()          # every tile of the interior, both strips and the corner were written
    rng = np.random.default_rng(M + K)
    x = rng.standard_normal(N)

    def matvec(t, vec):
        out = np.empty(t.shape[0])
        for r0 in range(0, t.shape[0], 2048):
            out[r0:r0 + 2048] = t[r0:r0 + 2048].cpu().numpy().astype(np.float64) @ vec
        return out
    want = matvec(a, matvec(b, x))
    got = matvec(c, x)
    rel = np.abs(got - want).max() / np.abs(want).max()
    assert rel < 2e-3, rel
    # sampled rows: interior, the last interior tile row, the bottom strip (when M % 256 == 128); every row crosses the right strip
    rows = [0, 255, 256, M // 2 + 1, (M // 256) * 256 - 1, M - 128, M - 1]
    truth = oracle.hgemm(a[rows].contiguous(), b, len(rows), N, K, 0, "f32")
    ok, mx, _ = tol.hgemm_close(c[rows].float().cpu().numpy(), truth, K)
    assert ok, mx
    capi.vendor_init()
    try:
        cv = torch.empty_like(c)
        capi.hgemm_vendor(a, bb, cv, lay)
        torch.cuda.synchronize()
        ulp = torch.clamp(cv.float().abs(), min=64.0) * 2.0 ** -10
        assert ((c.float() - cv.float()).abs() <= ulp).all()
    finally:
        capi.vendor_destroy()


@pytest.mark.parametrize("layout", ["nn", "tn"])
def test_hgemm_tail_split_border_and_k_tail_together(oracle, layout):
    """4480 x 4480 x 4128: 17 x 17 = 289 interior tiles on 256 CUs (33 in a ragged last wave -> 132 quadrant blocks on the 128-tile kernel),
    128-wide right AND bottom strips (35 + 34 blocks in the same launch), K % 64 == 32 in every kernel, split-K over all of those blocks —
    every special case of lc_abi.hip launch_mfma256 in one problem."""
    capi = _capi()
    M = N = 4480
    K = 4128
    lay = capi.LAYOUT_NN if layout == "nn" else capi.LAYOUT_TN
    torch.manual_seed(4480)
    a = torch.randn(M, K, dtype=torch.half, device="cuda")
    b = torch.randn(K, N, dtype=torch.half, device="cuda")
    bb = host.as_col_major(b) if lay == capi.LAYOUT_TN else b
    assert capi.hgemm_kernel_name(M, N, K, lay).startswith("hgemm_w4y_kernel")
    outs = {}
    for ks in (0, 1, 3):
        capi.tune("hgemm_splitk", ks)
        try:
            c = torch.full((M, N), float("nan"), dtype=torch.half, device="cuda")
            capi.hgemm(a, bb, c, layout=lay, variant=capi.HGEMM_AUTO, swizzle_stride=1024)
            torch.cuda.synchronize()
        finally:
            capi.tune("hgemm_splitk", 0)
        assert torch.isfinite(c).all(), ks
        outs[ks] = c
    rows = [0, 255, 256, 2241, 4351, 4352, 4479]
    truth = oracle.hgemm(a[rows].contiguous(), b, len(rows), N, K, 0, "f32")
    for ks, c in outs.items():
        ok, mx, _ = tol.hgemm_close(c[rows].float().cpu().numpy(), truth, K)
        assert ok, (ks, mx)
    capi.vendor_init()
    try:
        cv = torch.empty_like(outs[0])
        capi.hgemm_vendor(a, bb, cv, lay)
        torch.cuda.synchronize()
        ulp = torch.clamp(cv.float().abs(), min=64.0) * 2.0 ** -10
        for ks, c in outs.items():
            assert ((c.float() - cv.float()).abs() <= ulp).all(), ks
    finally:
        capi.vendor_destroy()


def test_hgemm_random_reference_legal_shapes_against_the_vendor_gemm():
    """Twelve random shapes with M, N multiples of 128 and K multiples of 32 (the reference's legality rule), LC_HGEMM_AUTO, NN and TN,
    against hipBLASLt on the same operands to fp16 rounding: whatever mix of flagship kernel, border strips, tail quadrants, split-K,
    four- / eight-wave 128-tile kernel the dispatcher picks."""
    capi = _capi()
    rng = np.random.default_rng(128)
    capi.vendor_init()
    try:
        for _ in range(12):
            M, N = (int(x) * 128 for x in rng.integers(2, 40, 2))
            K = int(rng.integers(2, 70)) * 32
            for lay in (capi.LAYOUT_NN, capi.LAYOUT_TN):
                torch.manual_seed(M + N + K)
                a = torch.randn(M, K, dtype=torch.half, device="cuda")
                b = torch.randn(K, N, dtype=torch.half, device="cuda")
                bb = host.as_col_major(b) if lay == capi.LAYOUT_TN else b
                c = torch.full((M, N), float("nan"), dtype=torch.half, device="cuda")
                cv = torch.empty_like(c)
                capi.hgemm(a, bb, c, layout=lay, variant=capi.HGEMM_AUTO, swizzle_stride=512)
                capi.hgemm_vendor(a, bb, cv, lay)
                torch.cuda.synchronize()
                ulp = torch.clamp(cv.float().abs(), min=32.0) * 2.0 ** -10
                bad = ((c.float() - cv.float()).abs() > ulp).sum().item()
                assert bad == 0, (M, N, K, lay, capi.hgemm_kernel_name(M, N, K, lay), bad)
    finally:
        capi.vendor_destroy()


def test_d256_full_size(oracle):
    """(1,48,8192,256) through the tiling-QKV entry and its shared-QKV stage-1 sibling (flash_attn_mma_share_qkv.cu:872-921: d = 256
    only with stages = 1)."""
    capi = _capi()
    B, H, N, D = 1, 48, 8192, 256
    torch.manual_seed(256)
    q = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    k = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    v = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    k[0, 47, 7000] = 3.0 * q[0, 47, 33]        # a late spike: forces whatever rescale path the kernel has
    v[0, 47, 7000] = 5.0
    o = torch.full_like(q, float("nan"))
    capi.attn_call("flash_attn_mma_stages_split_q_tiling_qkv", q, k, v, o, 2)
    torch.cuda.synchronize()
    assert torch.isfinite(o).all()
    rows = _rows_for(N, 256, extra=[33, 127, 128])
    _sampled_rows_check(oracle, q, k, v, o, [(0, 0), (0, 23)], rows)
    _sampled_rows_check(oracle, q, k, v, o, [(0, 47)], rows, rtol=tol.ATTN_RTOL_SPIKE)      # the spiked head
    o2 = torch.full_like(q, float("nan"))
    capi.attn_call("flash_attn_mma_stages_split_q_shared_qkv", q, k, v, o2, 1)
    torch.cuda.synchronize()
    assert torch.equal(o, o2)
    vc = torch.full_like(v, 0.625)
    capi.attn_call("flash_attn_mma_stages_split_q_tiling_qkv", q, k, vc, o, 2)
    torch.cuda.synchronize()
    assert (o.float() - 0.625).abs().max().item() < 1e-3


def test_d256_bf16_full_size(oracle):
    """Round-4 verdict (weak #1): bench.py times bf16 at (1,48,8192,256) on attn_fwd_bigd7_kernel<true,false>; its parity used to stop at
    N = 1024.  The bench shape itself: sampled rows x all keys against the bf16 oracle, a late spike head, one perturbed KV tile (first /
    ring wrap / last) must move the output, a constant V must come back exactly."""
    capi = _capi()
    B, H, N, D = 1, 48, 8192, 256
    torch.manual_seed(2560)
    q = torch.randn(B, H, N, D, device="cuda").to(torch.bfloat16)
    k = torch.randn(B, H, N, D, device="cuda").to(torch.bfloat16)
    v = torch.randn(B, H, N, D, device="cuda").to(torch.bfloat16)
    k[0, 47, 7000] = (3.0 * q[0, 47, 33].float()).to(torch.bfloat16)
    v[0, 47, 7000] = 5.0
    assert capi.attn_kernel_name(N, D, False, True, bh=B * H) == "attn_fwd_bigd7_kernel<true,false>"
    o = torch.full_like(q, float("nan"))
    capi.attn_fwd_bf16(q, k, v, o)
    torch.cuda.synchronize()
    assert torch.isfinite(o.float()).all()
    rows = _rows_for(N, 256, extra=[33, 127, 128])
    _sampled_rows_check(oracle, q, k, v, o, [(0, 0), (0, 23)], rows, bf16=True)
    _sampled_rows_check(oracle, q, k, v, o, [(0, 47)], rows, bf16=True, rtol=2.0 ** -6)      # the spiked head (bf16 scores: 8 bits)
    outs = []
    for t in (0, 4, N // 64 - 1):
        k2 = k.clone()
        k2[:, :2, 64 * t:64 * t + 64] = (k2[:, :2, 64 * t:64 * t + 64].float() * 1.25).to(torch.bfloat16)
        o2 = torch.full_like(q, float("nan"))
        capi.attn_fwd_bf16(q, k2, v, o2)
        torch.cuda.synchronize()
        _sampled_rows_check(oracle, q, k2, v, o2, [(0, 0), (0, 1)], rows, bf16=True)
        outs.append(o2[:, :2].clone())
        assert torch.equal(o2[:, 2:], o[:, 2:])           # the other heads never see the perturbation
    assert not torch.equal(outs[0], outs[1]) and not torch.equal(outs[1], outs[2])
    vc = torch.full_like(v, 0.625)
    capi.attn_fwd_bf16(q, k, vc, o)
    torch.cuda.synchronize()
    assert (o.float() - 0.625).abs().max().item() < 4e-3


@pytest.mark.parametrize("layout", ["nn", "tn"])
@pytest.mark.parametrize("variant,shape", [("mfma128", (4224, 4352, 4128)), ("generic", (4100, 4090, 4100)), ("generic", (8192, 136, 8204)),
                                           ("edge", (3000, 3016, 200)), ("edge", (8192, 136, 232)), ("kpad", (8192, 136, 8200)), ("ragged", (8200, 8264, 4128)), ("kpad", (4100, 4088, 4104)), ("kpad", (8192, 8192, 8200))])
def test_mid_size_kernels_at_the_sizes_they_serve(oracle, variant, shape, layout):
    """Round-4 verdict (weak #2): hgemm_mfma128_kernel / hgemm_generic_kernel had parity up to ~1000^3 only, while LC_HGEMM_AUTO routes
    4000-class problems to them (128-multiples with a small interior, K % 32 != 0, ragged M / N).  Late round 6: hgemm_edge_kernel (16-byte
    chunks) takes the ragged shapes with K % 8 == 0 (NN: N % 8 == 0), hgemm_generic_kernel what is left."""
    capi = _capi()
    M, N, K = shape
    lay = capi.LAYOUT_NN if layout == "nn" else capi.LAYOUT_TN
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, dtype=torch.half, device="cuda")
    b = torch.randn(K, N, dtype=torch.half, device="cuda")
    bb = host.as_col_major(b) if lay == capi.LAYOUT_TN else b
    var = {"mfma128": capi.HGEMM_MFMA128, "generic": capi.HGEMM_GENERIC, "edge": capi.HGEMM_EDGE, "ragged": capi.HGEMM_AUTO, "kpad": capi.HGEMM_AUTO}[variant]
    if variant == "kpad":     # K % 32 != 0 on a large problem: zero-padded operand copies in the workspace + the tuned kernels on the padded K (exactly the same sums)
        assert capi.hgemm_kernel_name(M, N, K, lay).startswith("hgemm_pad_copy_kernel + hgemm_w4y_kernel" if N > 1000 else "hgemm_pad_copy_kernel + hgemm_mid_edge")
    elif variant == "ragged":   # what LC_HGEMM_AUTO launches on a large ragged shape with K % 32 == 0: interior on the flagship kernel (K % 64 == 32: its half step) + border
        assert capi.hgemm_kernel_name(M, N, K, lay).startswith("hgemm_w4y_kernel") and "+ hgemm_mid_edge_kernel" in capi.hgemm_kernel_name(M, N, K, lay)
    else:
        assert capi.hgemm_kernel_name(M, N, K, lay, var).startswith(f"hgemm_{variant}_kernel")
    if variant in ("generic", "edge"):
        assert capi.hgemm_kernel_name(M, N, K, lay).startswith(f"hgemm_{variant}_kernel")   # what AUTO launches here (K % 8 != 0: element-wise)
    c = torch.full((M, N), float("nan"), dtype=torch.half, device="cuda")
    capi.hgemm(a, bb, c, layout=lay, variant=var, swizzle_stride=1024)
    torch.cuda.synchronize()
    assert torch.isfinite(c).all()
    rows = sorted({0, 1, 127, 128, M // 2 + 3, M - 129, M - 2, M - 1})
    truth = oracle.hgemm(a[rows].contiguous(), b, len(rows), N, K, 0, "f32")
    ok, mx, _ = tol.hgemm_close(c[rows].float().cpu().numpy(), truth, K)
    assert ok, mx
    capi.vendor_init()
    try:
        cv = torch.empty_like(c)
        capi.hgemm_vendor(a, bb, cv, lay)
        torch.cuda.synchronize()
        ulp = torch.clamp(cv.float().abs(), min=64.0) * 2.0 ** -10
        assert ((c.float() - cv.float()).abs() <= ulp).all()
    finally:
        capi.vendor_destroy()


@pytest.mark.parametrize("entry", ["flash_attn_mma_stages_split_q_shared_qkv_swizzle_qkv",
                                   "flash_attn_mma_stages_split_q_shared_kv_swizzle_qkv",
                                   "flash_attn_mma_stages_split_q_tiling_qk_swizzle_qkv"])
def test_v_transposed_entries_config3(oracle, entry):
    """The three *_swizzle_qkv entries take V as [B,H,D,N] (flash_attn_mma.py:441-442,716; share_qkv_swizzle_qkv.cu:961-968) —
    at config 3's shape, against the oracle reading the same transposed tensor, and against the [B,H,N,D] sibling."""
    capi = _capi()
    B, H, N, D = 4, 32, 4096, 128
    torch.manual_seed(3)
    q = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    k = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    v = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    k[3, 31, 3000] = 3.0 * q[3, 31, 33]
    v[3, 31, 3000] = 5.0
    tv = v.transpose(-2, -1).contiguous()
    o = torch.full_like(q, float("nan"))
    capi.attn_call(entry, q, k, tv, o, 2)
    torch.cuda.synchronize()
    assert torch.isfinite(o).all()
    rows = _rows_for(N, 3, extra=[33])
    _sampled_rows_check(oracle, q, k, tv, o, [(0, 0), (1, 13), (2, 7)], rows, vt=True)
    _sampled_rows_check(oracle, q, k, tv, o, [(3, 31)], rows, vt=True, rtol=tol.ATTN_RTOL_SPIKE)      # the spiked head
    os_ = torch.full_like(q, float("nan"))
    capi.attn_call(entry.replace("_swizzle_qkv", ""), q, k, v, os_, 2)
    torch.cuda.synchronize()
    # same products; the P.V operand order inside an MFMA may differ between the two V layouts: fp16 rounding, not more
    assert (o.float() - os_.float()).abs().max().item() <= 2.0 ** -10 * max(1.0, os_.float().abs().max().item())
    # V = const => O = const on every row
    vc = torch.full_like(tv, -0.375)
    capi.attn_call(entry, q, k, vc, o, 2)
    torch.cuda.synchronize()
    assert (o.float() + 0.375).abs().max().item() < 1e-3


def test_d256_v_transposed_full_size(oracle):
    """(1,48,8192,256) with V as [B,H,D,N] — the largest head dim the reference's *_swizzle_qkv entries take
    (flash_attn_mma_share_qkv_swizzle_qkv.cu:961-1010: d = 256 with stages = 1; tiling_qk_swizzle_qkv: d <= 256) — runs
    attn_fwd_bigd7_kernel<false,true> (round 3 / knob 3: attn_fwd_bigd2_kernel<256,false,true>): K rows fed in the order that makes a
    lane's P slots contiguous kv, Vᵀ fragments one ds_read_b128 each.  Against the oracle reading the same transposed tensor and against the [B,H,N,D] sibling."""
    capi = _capi()
    B, H, N, D = 1, 48, 8192, 256
    torch.manual_seed(2560)
    q = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    k = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    v = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    k[0, 47, 7000] = 3.0 * q[0, 47, 33]
    v[0, 47, 7000] = 5.0
    tv = v.transpose(-2, -1).contiguous()
    assert capi.attn_kernel_name(N, D, True) == "attn_fwd_bigd7_kernel<false,true>"
    rows = _rows_for(N, 2560, extra=[33, 127, 128])
    o = torch.full_like(q, float("nan"))
    for entry, st in (("flash_attn_mma_stages_split_q_tiling_qk_swizzle_qkv", 2), ("flash_attn_mma_stages_split_q_shared_qkv_swizzle_qkv", 1)):
        o.fill_(float("nan"))
        capi.attn_call(entry, q, k, tv, o, st)
        torch.cuda.synchronize()
        assert torch.isfinite(o).all()
        _sampled_rows_check(oracle, q, k, tv, o, [(0, 0), (0, 23)], rows, vt=True)
        _sampled_rows_check(oracle, q, k, tv, o, [(0, 47)], rows, vt=True, rtol=tol.ATTN_RTOL_SPIKE)
    os_ = torch.full_like(q, float("nan"))
    capi.attn_call("flash_attn_mma_stages_split_q_tiling_qk", q, k, v, os_, 2)
    torch.cuda.synchronize()
    assert (o.float() - os_.float()).abs().max().item() <= 2.0 ** -10 * max(1.0, os_.float().abs().max().item())
    o2 = torch.full_like(q, float("nan"))
    capi.attn_call("flash_attn_mma_stages_split_q_tiling_qk_swizzle_qkv", q, k, tv, o2, 2)
    torch.cuda.synchronize()
    assert torch.equal(o, o2)                      # launch to launch
    vc = torch.full_like(tv, 0.625)
    capi.attn_call("flash_attn_mma_stages_split_q_tiling_qk_swizzle_qkv", q, k, vc, o, 2)
    torch.cuda.synchronize()
    assert (o.float() - 0.625).abs().max().item() < 1e-3


@pytest.mark.parametrize("N", [2048, 8192])
def test_d1024_full_width(oracle, N):
    """D = 1024 (flash_attn_mma_tiling_qkv.cu:904-910,931-937 and tiling_qk dispatch up to 1024) at N >= 2048 — round 3 tested N = 128
    only."""
    capi = _capi()
    B, H, D = 1, 4 if N == 8192 else 8, 1024
    torch.manual_seed(1024 + N)
    q = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    k = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    v = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    k[0, H - 1, N - 100] = 3.0 * q[0, H - 1, 33]
    v[0, H - 1, N - 100] = 5.0
    rows = _rows_for(N, 1024, extra=[33, 127, 128])
    o = torch.full_like(q, float("nan"))
    for entry in ("flash_attn_mma_stages_split_q_tiling_qkv", "flash_attn_mma_stages_split_q_tiling_qk"):
        o.fill_(float("nan"))
        capi.attn_call(entry, q, k, v, o, 2)
        torch.cuda.synchronize()
        assert torch.isfinite(o).all()
        _sampled_rows_check(oracle, q, k, v, o, [(0, 0)], rows)
        _sampled_rows_check(oracle, q, k, v, o, [(0, H - 1)], rows, rtol=tol.ATTN_RTOL_SPIKE)      # the spiked head
    assert capi.attn_kernel_name(N, D) == "attn_fwd_bigd4_kernel<8>"
    # the pair kernel against the independently written round-1 column-split kernel: same products, other summation order
    o1 = torch.full_like(q, float("nan"))
    capi.tune("attn_d512", 1)
    try:
        capi.attn_call("flash_attn_mma_stages_split_q_tiling_qkv", q, k, v, o1, 2)
        torch.cuda.synchronize()
    finally:
        capi.tune("attn_d512", 0)
    assert (o.float() - o1.float()).abs().max().item() <= 2.0 ** -9 * max(1.0, o1.float().abs().max().item())
    o2 = torch.full_like(q, float("nan"))
    capi.attn_call("flash_attn_mma_stages_split_q_tiling_qkv", q, k, v, o2, 2)
    torch.cuda.synchronize()
    assert torch.equal(o, o2)                      # launch to launch (both members of a pair hold the same S: a + b == b + a)
    vc = torch.full_like(v, 0.875)
    capi.attn_call("flash_attn_mma_stages_split_q_tiling_qkv", q, k, vc, o, 2)
    torch.cuda.synchronize()
    assert (o.float() - 0.875).abs().max().item() < 1e-3
