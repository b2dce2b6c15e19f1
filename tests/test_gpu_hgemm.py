"""GPU parity tests of the HGEMM path, all through the C-ABI, against the CPU oracle / golden fixtures.
Sizes: oracle-checked up to 1024^3 (config 1's size); BASELINE config 2 (8192^3) is checked with
size-independent properties (C·x == A·(B·x), sampled rows against the exact oracle, agreement with the
vendor GEMM)."""
import numpy as np
import pytest
import torch

from leetcuda_amd import host
from tests import tol

pytestmark = pytest.mark.gpu

VARIANTS = {"mfma256": 1, "generic": 3, "pingpong2": 4, "w4b": 9, "w4c": 10, "w4x": 12, "w4y": 13}   # lc_hgemm_variant (lc_abi.h)


def _capi():
    from leetcuda_amd import capi
    capi.load()
    return capi


def _run(capi, a, b, layout, variant, stride=1):
    M, K = a.shape
    N = b.shape[1]
    bb = host.as_col_major(b) if layout == capi.LAYOUT_TN else b
    c = torch.full((M, N), float("nan"), dtype=torch.half, device="cuda")
    capi.hgemm(a, bb, c, layout=layout, variant=variant, swizzle_stride=stride)
    torch.cuda.synchronize()
    return c, bb


def _check(oracle, capi, a, b, c, layout, amp=1.0):
    M, K = a.shape
    N = b.shape[1]
    truth = oracle.hgemm(a, b.contiguous(), M, N, K, 0, "f32")
    out = c.float().cpu().numpy()
    assert np.isfinite(out).all()
    ok, mx, ex = tol.hgemm_close(out, truth, K, amp)
    assert ok, f"max abs err {mx}, worst excess {ex}"
    exact = oracle.hgemm(a, b.contiguous(), M, N, K, 0, "exact")
    same = (c.cpu().numpy().view(np.uint16) == exact.view(np.uint16)).mean()
    assert same > 0.97, f"only {same:.4f} of outputs are the correctly rounded fp16 result"
    return mx


@pytest.mark.parametrize("layout", ["nn", "tn"])
@pytest.mark.parametrize("variant", ["mfma256", "pingpong2", "w4b", "w4c", "w4x", "w4y"])
@pytest.mark.parametrize("shape", [(256, 256, 64), (256, 512, 128), (512, 256, 448), (1024, 1024, 1024)])
def test_tuned_kernels_vs_oracle(oracle, variant, layout, shape):
    capi = _capi()
    M, N, K = shape
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, dtype=torch.half, device="cuda")
    b = torch.randn(K, N, dtype=torch.half, device="cuda")
    lay = capi.LAYOUT_NN if layout == "nn" else capi.LAYOUT_TN
    for stride in (1, 512):
        c, _ = _run(capi, a, b, lay, VARIANTS[variant], stride)
        _check(oracle, capi, a, b, c, lay)


@pytest.mark.parametrize("layout", ["nn", "tn"])
@pytest.mark.parametrize("shape", [(128, 128, 64), (384, 128, 192), (128, 640, 128), (896, 1152, 512)])
def test_mfma128_reference_tile_shapes(oracle, layout, shape):
    """M, N multiples of 128 (the reference's own 128x128 tiles), not of 256: the 128-tile kernel, explicitly
    and through LC_HGEMM_AUTO."""
    capi = _capi()
    M, N, K = shape
    torch.manual_seed(M * 5 + N + K)
    a = torch.randn(M, K, dtype=torch.half, device="cuda")
    b = torch.randn(K, N, dtype=torch.half, device="cuda")
    lay = capi.LAYOUT_NN if layout == "nn" else capi.LAYOUT_TN
    for var, stride in ((capi.HGEMM_MFMA128, 1), (capi.HGEMM_AUTO, 256)):
        c, _ = _run(capi, a, b, lay, var, stride)
        _check(oracle, capi, a, b, c, lay)


@pytest.mark.parametrize("layout", ["nn", "tn"])
@pytest.mark.parametrize("shape", [(256, 256, 96), (512, 256, 160), (256, 512, 1056),           # K % 64 == 32 on the 256 tile
                                   (384, 384, 128), (640, 256, 64), (256, 896, 192),             # 128-wide border strips
                                   (384, 640, 96), (896, 1152, 544),                             # both
                                   (384, 640, 2080), (640, 384, 4096)])                          # long K: the border blocks run split-K
def test_flagship_kernel_on_the_reference_legal_shapes(oracle, layout, shape):
    """Round-4 verdict (missing #1): the reference's kernels are legal on M, N multiples of 128 and K multiples of 32
    (hgemm_mma_stage.cu:650,675-676).  hgemm_w4y_kernel now takes them: a half K-step behind its generated loop (fragments straight
    from global memory) and the 128-wide right / bottom border strips on the 128-tile kernel in a second launch."""
    capi = _capi()
    M, N, K = shape
    torch.manual_seed(M * 11 + N * 3 + K)
    a = torch.randn(M, K, dtype=torch.half, device="cuda")
    b = torch.randn(K, N, dtype=torch.half, device="cuda")
    lay = capi.LAYOUT_NN if layout == "nn" else capi.LAYOUT_TN
    assert capi.hgemm_kernel_name(M, N, K, lay, capi.HGEMM_MFMA256W4Y).startswith("hgemm_w4y_kernel")
    for stride in (1, 256):
        c, _ = _run(capi, a, b, lay, capi.HGEMM_MFMA256W4Y, stride)
        _check(oracle, capi, a, b, c, lay)
    # the other 256-tile kernels keep their own contract (256-multiples, K % 64 == 0): a clean error, never a wrong answer
    c = torch.zeros(M, N, dtype=torch.half, device="cuda")
    bb = host.as_col_major(b) if lay == capi.LAYOUT_TN else b
    with pytest.raises(capi.LcError, match="Tensor size mismatch"):
        capi.hgemm(a, bb, c, layout=lay, variant=VARIANTS["w4x"] if layout == "tn" else VARIANTS["w4c"])


@pytest.mark.parametrize("layout", ["nn", "tn"])
def test_border_strips_split_k_factors(oracle, layout):
    """lc_tune_set "hgemm_splitk": the 128-tile blocks of the border strips walk 1 / ks of K each and a second kernel adds their fp32
    partials (auto: ks = 8 here — 5 blocks on 256 CUs, 65 K tiles).  Every factor against the oracle, K % 64 == 32 in the last range;
    the factors agree with each other to fp16 rounding (only the fp32 summation order differs)."""
    capi = _capi()
    M, N, K = 384, 384, 4192
    torch.manual_seed(4192)
    a = torch.randn(M, K, dtype=torch.half, device="cuda")
    b = torch.randn(K, N, dtype=torch.half, device="cuda")
    lay = capi.LAYOUT_NN if layout == "nn" else capi.LAYOUT_TN
    outs = {}
    for ks in (1, 2, 3, 5, 8, 0):
        capi.tune("hgemm_splitk", ks)
        try:
            c, _ = _run(capi, a, b, lay, capi.HGEMM_MFMA256W4Y, 256)
        finally:
            capi.tune("hgemm_splitk", 0)
        _check(oracle, capi, a, b, c, lay)
        outs[ks] = c
    assert torch.equal(outs[0], outs[8])                      # auto = 8 on this shape
    assert torch.equal(outs[1][:256, :256], outs[8][:256, :256])        # the interior tile never depends on the knob
    ulp = torch.clamp(outs[1].float().abs(), min=32.0) * 2.0 ** -10
    for ks in (2, 3, 5, 8):
        assert ((outs[ks].float() - outs[1].float()).abs() <= ulp).all(), ks


def test_border_strips_inside_graph_capture(oracle):
    """The split-K of the border blocks needs the cached workspace (hipMalloc on first use): while a stream is being captured the launcher
    runs the blocks unsplit instead, so a captured 128-multiple GEMM is plain kernel nodes, and the replay computes the same product."""
    capi = _capi()
    M, N, K = 640, 384, 4192
    torch.manual_seed(640)
    a = torch.randn(M, K, dtype=torch.half, device="cuda")
    b = torch.randn(K, N, dtype=torch.half, device="cuda")
    bb = host.as_col_major(b)
    c = torch.zeros(M, N, dtype=torch.half, device="cuda")
    for ks in (0, 1):                      # warm-up of both forms outside capture (kernel attributes are set on first use)
        capi.tune("hgemm_splitk", ks)
        try:
            capi.hgemm(a, bb, c, layout=capi.LAYOUT_TN, variant=capi.HGEMM_MFMA256W4Y, swizzle_stride=256)
        finally:
            capi.tune("hgemm_splitk", 0)
    torch.cuda.synchronize()
    c_eager = c.clone()
    c.zero_()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        capi.hgemm(a, bb, c, layout=capi.LAYOUT_TN, variant=capi.HGEMM_MFMA256W4Y, swizzle_stride=256)
    g.replay()
    torch.cuda.synchronize()
    _check(oracle, capi, a, b, c, capi.LAYOUT_TN)
    ulp = torch.clamp(c_eager.float().abs(), min=32.0) * 2.0 ** -10
    assert ((c.float() - c_eager.float()).abs() <= ulp).all()


@pytest.mark.parametrize("layout", ["nn", "tn"])
@pytest.mark.parametrize("shape", [(128, 128, 96), (384, 128, 160), (256, 384, 224), (128, 256, 1056)])
def test_mfma128_half_k_step(oracle, layout, shape):
    """K % 64 == 32 on the 128-tile kernel (explicitly and as LC_HGEMM_AUTO's choice for small grids), in its four-wave and eight-wave
    forms (lc_tune_set "hgemm_128w": the eight-wave form splits every K tile's two k-steps over two groups of four waves)."""
    capi = _capi()
    M, N, K = shape
    torch.manual_seed(M * 13 + N + K)
    a = torch.randn(M, K, dtype=torch.half, device="cuda")
    b = torch.randn(K, N, dtype=torch.half, device="cuda")
    lay = capi.LAYOUT_NN if layout == "nn" else capi.LAYOUT_TN
    assert capi.hgemm_kernel_name(M, N, K, lay).startswith("hgemm_mfma128_kernel")
    for w in (1, 2, 0):
        capi.tune("hgemm_128w", w)
        try:
            assert capi.hgemm_kernel_name(M, N, K, lay).endswith(f",{w or 2}>")
            for var in (capi.HGEMM_MFMA128, capi.HGEMM_AUTO):
                c, _ = _run(capi, a, b, lay, var, 256)
                _check(oracle, capi, a, b, c, lay)
        finally:
            capi.tune("hgemm_128w", 0)


@pytest.mark.parametrize("layout", ["nn", "tn"])
def test_mfma128_eight_waves_at_the_size_it_serves(oracle, layout):
    """1536^3 = 144 blocks of 128 x 128 on 256 CUs: the eight-wave 128-tile kernel (round 5's LC_HGEMM_AUTO choice at this size; since round 6
    the mid-size kernel takes it — "hgemm_mid" = 1 restores the old rule, which still serves grids of <= 48 blocks) against the oracle, the
    four-wave form (same products, the two k-steps of a K tile summed in a different order: fp16-rounding agreement) and the identity trick."""
    capi = _capi()
    capi.tune("hgemm_mid", 1)
    try:
        _eight_waves_body(oracle, capi, layout)
    finally:
        capi.tune("hgemm_mid", 0)


def _eight_waves_body(oracle, capi, layout):
    n = 1536
    torch.manual_seed(1536)
    a = torch.randn(n, n, dtype=torch.half, device="cuda")
    b = torch.randn(n, n, dtype=torch.half, device="cuda")
    lay = capi.LAYOUT_NN if layout == "nn" else capi.LAYOUT_TN
    nn = "true" if layout == "nn" else "false"
    assert capi.hgemm_kernel_name(n, n, n, lay) == f"hgemm_mfma128_kernel<{nn},2>"
    c8, _ = _run(capi, a, b, lay, capi.HGEMM_AUTO, 1024)
    rows = [0, 63, 64, 127, 128, 1025, n - 1]
    truth = oracle.hgemm(a[rows].contiguous(), b, len(rows), n, n, 0, "f32")
    ok, mx, _ = tol.hgemm_close(c8[rows].float().cpu().numpy(), truth, n)
    assert ok, mx
    capi.tune("hgemm_128w", 1)
    try:
        c4, _ = _run(capi, a, b, lay, capi.HGEMM_AUTO, 1024)
    finally:
        capi.tune("hgemm_128w", 0)
    ulp = torch.clamp(c4.float().abs(), min=32.0) * 2.0 ** -10
    assert ((c8.float() - c4.float()).abs() <= ulp).all()
    eye = torch.eye(n, dtype=torch.half, device="cuda")
    bq = (torch.arange(n * n, device="cuda").reshape(n, n) % 1021).half() / 4
    c, _ = _run(capi, eye, bq, lay, capi.HGEMM_AUTO, 1024)
    assert torch.equal(c, bq)


MID_COMBOS = [(lay, tmw, tnw, ns) for lay in ("tn", "nn") for tmw in (1, 2, 3) for tnw in ((2, 3) if lay == "tn" else (2,)) for ns in (2, 3)]


@pytest.mark.parametrize("layout,tmw,tnw,ns", MID_COMBOS)
def test_mid_kernel_every_tile_and_ring_depth(oracle, layout, tmw, tnw, ns):
    """hgemm_mid_kernel<B_KN, TMW, TNW, NS> (round 6: 64 TMW x 64 TNW tiles, NS ring slots, the hand-ordered k-loop rotated around a mid-tile
    barrier) through LC_HGEMM_MID with the tile / depth knobs: against the oracle on shapes whose K walk is SHORTER than the ring (K = 64 ...
    128: the prologue's clamped requests, the tail loop's wait counts), equal to it, longer (main loop + tail), and with the K % 64 == 32
    half step; several tiles in M and N, both block maps; bit-equal to hgemm_mfma128_kernel (same products, k ascending into one fp32
    accumulator) where that kernel tiles the shape."""
    capi = _capi()
    lay = capi.LAYOUT_NN if layout == "nn" else capi.LAYOUT_TN
    tm, tn = 64 * tmw, 64 * tnw
    capi.tune("hgemm_mid", 10 * tmw + tnw)
    capi.tune("hgemm_mid_ns", ns)
    try:
        nnn = "true" if layout == "nn" else "false"
        for (M, N, K) in [(tm, tn, 64), (tm, tn, 128), (2 * tm, 2 * tn, 192), (tm, tn, 256), (3 * tm, 3 * tn, 320), (2 * tm, tn, 96),
                          (tm, 2 * tn, 160), (2 * tm, 2 * tn, 1056), (5 * tm, 3 * tn, 512)]:
            assert capi.hgemm_kernel_name(M, N, K, lay, capi.HGEMM_MID) == f"hgemm_mid_kernel<{nnn},{tmw},{tnw},{ns}>"
            torch.manual_seed(M + N + K + tnw)
            a = torch.randn(M, K, dtype=torch.half, device="cuda")
            b = torch.randn(K, N, dtype=torch.half, device="cuda")
            for stride in (1, 2 * tn):
                c, _ = _run(capi, a, b, lay, capi.HGEMM_MID, stride)
                _check(oracle, capi, a, b, c, lay)
            if M % 128 == 0 and N % 128 == 0:
                capi.tune("hgemm_128w", 1)      # (the eight-wave form sums the two k-steps of a K tile apart)
                try:
                    c128, _ = _run(capi, a, b, lay, capi.HGEMM_MFMA128, 1)
                finally:
                    capi.tune("hgemm_128w", 0)
                assert torch.equal(c, c128), (M, N, K)
        # the identity trick: a wrong fragment / tile / transpose shows as a permutation
        n = 3 * 128 if tnw == 3 or tmw == 3 else 256
        eye = torch.eye(n, dtype=torch.half, device="cuda")
        bq = (torch.arange(n * n, device="cuda").reshape(n, n) % 1021).half() / 4
        c, _ = _run(capi, eye, bq, lay, capi.HGEMM_MID, 1)
        assert torch.equal(c, bq)
        # a shape the forced tile cannot divide is refused, never mis-computed
        bad_n = tn + 64
        with pytest.raises(RuntimeError):
            bb = torch.zeros(bad_n, 64, dtype=torch.half, device="cuda") if lay == capi.LAYOUT_TN else torch.zeros(64, bad_n, dtype=torch.half, device="cuda")
            capi.hgemm(torch.zeros(tm, 64, dtype=torch.half, device="cuda"), bb, torch.zeros(tm, bad_n, dtype=torch.half, device="cuda"),
                       layout=lay, variant=capi.HGEMM_MID)
    finally:
        capi.tune("hgemm_mid", 0)
        capi.tune("hgemm_mid_ns", 0)


@pytest.mark.parametrize("layout", ["nn", "tn"])
@pytest.mark.parametrize("n", [1280, 1536, 1792, 2048, 2304, 2560, 2816, 3072])
def test_mid_kernel_at_the_sizes_it_serves(oracle, layout, n):
    """The sizes of the reference's default sweep (hgemm.py:28-32: multiples of 256) that LC_HGEMM_AUTO hands to the mid-size kernel on a
    256-CU device: the tile the rule picks, sampled rows x full K against the oracle, C x = A (B x) in fp64, equality with the 128-tile
    kernel bit for bit, hipBLASLt within one fp16 ulp."""
    capi = _capi()
    lay = capi.LAYOUT_NN if layout == "nn" else capi.LAYOUT_TN
    name = capi.hgemm_kernel_name(n, n, n, lay)
    ncu = capi.device_check()
    if ncu == 256:
        want = {1792: "2,2,3", 2048: "2,2,3", 2304: "2,3,3" if layout == "tn" else "3,2,3", 2560: "2,2,2", 2816: "2,2,2",
                3072: "3,3,3" if layout == "tn" else None}.get(n)    # (3072 NN: no one-round tile, the 256-tile kernel keeps it)
        if want:
            assert name == f"hgemm_mid_kernel<{'true' if layout == 'nn' else 'false'},{want}>", name
    torch.manual_seed(n)
    a = torch.randn(n, n, dtype=torch.half, device="cuda")
    b = torch.randn(n, n, dtype=torch.half, device="cuda")
    stride = host.make_block_swizzle_stride(n, n)
    c, bb = _run(capi, a, b, lay, capi.HGEMM_AUTO, stride)
    rows = [0, 63, 64, 127, 128, n // 2 + 1, n - 129, n - 1]
    truth = oracle.hgemm(a[rows].contiguous(), b, len(rows), n, n, 0, "f32")
    ok, mx, _ = tol.hgemm_close(c[rows].float().cpu().numpy(), truth, n)
    assert ok, mx
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n)
    want = a.cpu().numpy().astype(np.float64) @ (b.cpu().numpy().astype(np.float64) @ x)
    got = c.cpu().numpy().astype(np.float64) @ x
    assert np.abs(got - want).max() / np.abs(want).max() < 2e-3
    capi.tune("hgemm_mid", 1)
    capi.tune("hgemm_128w", 1)
    try:
        c128, _ = _run(capi, a, b, lay, capi.HGEMM_MFMA128, stride)
    finally:
        capi.tune("hgemm_mid", 0)
        capi.tune("hgemm_128w", 0)
    if name.startswith("hgemm_mid_kernel"):
        assert torch.equal(c, c128)
    capi.vendor_init()
    try:
        cv = torch.empty_like(c)
        capi.hgemm_vendor(a, bb, cv, lay)
        torch.cuda.synchronize()
    finally:
        capi.vendor_destroy()
    ulp = torch.clamp(c.float().abs(), min=32.0) * 2.0 ** -10
    assert ((c.float() - cv.float()).abs() <= ulp).all()


@pytest.mark.parametrize("layout", ["nn", "tn"])
def test_mid_kernel_split_k_and_64_multiples(oracle, layout):
    """Late round 6.  (i) Split-K of the mid-size kernel (hgemm_mid_sk_kernel + hgemm_mid_reduce_kernel; fp32 partials in the stream's
    workspace): LC_HGEMM_AUTO picks it for one-round grids on at most half the CUs with a long K; every forced factor against the oracle
    incl. the K % 64 == 32 half step and K ranges of unequal length, equal to the unsplit launch to the rounding of differently grouped
    fp32 sums, bit-identical from run to run; one K range under graph capture (no workspace inside a graph).  (ii) M, N multiples of 64
    that are not multiples of 128 — not legal in the reference, hgemm_generic_kernel until now — run a mid-size tile wherever one divides
    the shape."""
    capi = _capi()
    lay = capi.LAYOUT_NN if layout == "nn" else capi.LAYOUT_TN
    nnn = "true" if layout == "nn" else "false"
    if capi.device_check() == 256:
        assert capi.hgemm_kernel_name(1024, 1024, 8192, lay) == f"hgemm_mid_sk_kernel<{nnn},1,3> x2"
    for (M, N, K) in ((512, 512, 8224), (256, 384, 4128), (1024, 1024, 8192)):
        torch.manual_seed(M + N + K)
        a = torch.randn(M, K, dtype=torch.half, device="cuda")
        b = torch.randn(K, N, dtype=torch.half, device="cuda")
        rows = sorted({0, 63, 64, M // 2 + 1, M - 1})
        truth = oracle.hgemm(a[rows].contiguous(), b, len(rows), N, K, 0, "f32")
        capi.tune("hgemm_mid_splitk", 1)
        try:
            c1, _ = _run(capi, a, b, lay, capi.HGEMM_MID, 1)
        finally:
            capi.tune("hgemm_mid_splitk", 0)
        outs = {}
        for tile in (12, 22):
            for ks in (0, 2, 3, 7, 8):
                capi.tune("hgemm_mid", tile)
                capi.tune("hgemm_mid_splitk", ks)
                try:
                    name = capi.hgemm_kernel_name(M, N, K, lay, capi.HGEMM_MID)
                    c, _ = _run(capi, a, b, lay, capi.HGEMM_MID, 1)
                    c2, _ = _run(capi, a, b, lay, capi.HGEMM_MID, 256)
                finally:
                    capi.tune("hgemm_mid", 0)
                    capi.tune("hgemm_mid_splitk", 0)
                if ks >= 2:
                    assert name.startswith("hgemm_mid_sk_kernel<") and name.endswith(f"> x{ks}"), name
                assert torch.equal(c, c2), (M, N, K, tile, ks)                 # another block map, another run: the same bits
                ok, mx, _ = tol.hgemm_close(c[rows].float().cpu().numpy(), truth, K)
                assert ok, (M, N, K, tile, ks, mx)
                ulp = torch.clamp(c1.float().abs(), min=32.0) * 2.0 ** -10
                assert ((c.float() - c1.float()).abs() <= ulp).all(), (M, N, K, tile, ks)
                outs[(tile, ks)] = c
        assert torch.equal(outs[(12, 8)], outs[(22, 8)])                       # the tile does not change a range's sum (same k order)
    # under graph capture: one K range, no workspace, same result as the unsplit launch
    M = N = 512
    K = 8192
    a = torch.randn(M, K, dtype=torch.half, device="cuda")
    b = torch.randn(K, N, dtype=torch.half, device="cuda")
    bb = host.as_col_major(b) if lay == capi.LAYOUT_TN else b
    capi.tune("hgemm_mid_splitk", 1)
    try:
        c1, _ = _run(capi, a, b, lay, capi.HGEMM_MID, 1)
    finally:
        capi.tune("hgemm_mid_splitk", 0)
    cg = torch.full((M, N), float("nan"), dtype=torch.half, device="cuda")
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        capi.hgemm(a, bb, cg, layout=lay, variant=capi.HGEMM_MID)             # (warm-up on the capture stream: split)
        torch.cuda.synchronize()
        cg.fill_(float("nan"))
        with torch.cuda.graph(g, stream=s):
            capi.hgemm(a, bb, cg, layout=lay, variant=capi.HGEMM_MID)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(cg, c1)
    # (ii) multiples of 64
    for (M, N, K) in ((1088, 1152, 512), (2880, 2880, 1056), (192, 8256, 320)):
        name = capi.hgemm_kernel_name(M, N, K, lay)
        if layout == "tn" or N % 128 == 0:
            assert name.startswith("hgemm_mid_kernel<"), (M, N, K, name)
        else:
            assert name.startswith("hgemm_mid_edge_kernel<"), (M, N, K, name)  # (NN has 128-column tiles only: clamped 128 x 128 tiles)
        torch.manual_seed(M + N)
        a = torch.randn(M, K, dtype=torch.half, device="cuda")
        b = torch.randn(K, N, dtype=torch.half, device="cuda")
        c, _ = _run(capi, a, b, lay, capi.HGEMM_AUTO, 1)
        rows = sorted(r for r in {0, 63, 64, 191, 192, M // 2 + 1, M - 65, M - 1} if 0 <= r < M)
        truth = oracle.hgemm(a[rows].contiguous(), b, len(rows), N, K, 0, "f32")
        ok, mx, _ = tol.hgemm_close(c[rows].float().cpu().numpy(), truth, K)
        assert ok and torch.isfinite(c).all(), (M, N, K, mx)
        cgen, _ = _run(capi, a, b, lay, capi.HGEMM_GENERIC, 1)
        ulp = torch.clamp(cgen.float().abs(), min=32.0) * 2.0 ** -10
        assert ((c.float() - cgen.float()).abs() <= ulp).all(), (M, N, K)


@pytest.mark.parametrize("layout", ["nn", "tn"])
@pytest.mark.parametrize("shape", [(3200, 3200, 96), (3072, 3456, 160), (3456, 3072, 128)])
def test_auto_routes_128_multiples_with_a_large_interior_to_the_flagship_kernel(oracle, layout, shape):
    """LC_HGEMM_AUTO and the reference's entry names: > 128 interior tiles of 256 x 256 -> hgemm_w4y_kernel + border strips (round 4:
    the whole problem fell to the 128-tile kernel, - 32 % at 8192-class sizes).  Rows of the bottom strip, columns of the right strip
    and the corner are all inside the full comparison."""
    capi = _capi()
    M, N, K = shape
    torch.manual_seed(M + 7 * N + K)
    a = torch.randn(M, K, dtype=torch.half, device="cuda")
    b = torch.randn(K, N, dtype=torch.half, device="cuda")
    lay = capi.LAYOUT_NN if layout == "nn" else capi.LAYOUT_TN
    assert capi.hgemm_kernel_name(M, N, K, lay).startswith("hgemm_w4y_kernel")
    c, bb = _run(capi, a, b, lay, capi.HGEMM_AUTO, 1024)
    _check(oracle, capi, a, b, c, lay)
    name = ("hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem" if layout == "nn" else "hgemm_mma_stages_block_swizzle_tn_cute")
    c2 = torch.full_like(c, float("nan"))
    capi.hgemm_call(name, a, bb, c2, 2, True, 1024)
    torch.cuda.synchronize()
    assert torch.equal(c, c2)


@pytest.mark.parametrize("layout", ["nn", "tn"])
@pytest.mark.parametrize("shape", [(64, 64, 64), (128, 128, 32), (100, 72, 50), (1, 1, 1), (257, 129, 65),
                                   (384, 640, 96)])
def test_generic_kernel_ragged_shapes(oracle, layout, shape):
    capi = _capi()
    M, N, K = shape
    torch.manual_seed(M * 7 + N * 3 + K)
    a = torch.randn(M, K, dtype=torch.half, device="cuda")
    b = torch.randn(K, N, dtype=torch.half, device="cuda")
    lay = capi.LAYOUT_NN if layout == "nn" else capi.LAYOUT_TN
    c, _ = _run(capi, a, b, lay, VARIANTS["generic"])
    _check(oracle, capi, a, b, c, lay)


@pytest.mark.parametrize("layout", ["nn", "tn"])
@pytest.mark.parametrize("shape", [(64, 64, 64), (128, 128, 32), (100, 72, 56), (1, 8, 8), (257, 136, 72), (384, 640, 96), (129, 1000, 40),
                                   (1000, 3000, 520), (130, 130, 64), (2880, 2944, 264)])
def test_edge_kernel_ragged_shapes(oracle, layout, shape):
    """Late round 6: hgemm_edge_kernel (16-byte chunks, whole-chunk predication) is what LC_HGEMM_AUTO runs where no tiled kernel divides
    the shape (any M, N; K % 8 == 0; NN: N % 8 == 0).  Against the oracle, against the element-wise hgemm_generic_kernel (other fp32 order:
    one output ulp), bit-identical from run to run, NaN canaries around C untouched; (130, 130, 64) is TN-only (N % 8 != 0)."""
    capi = _capi()
    M, N, K = shape
    lay = capi.LAYOUT_NN if layout == "nn" else capi.LAYOUT_TN
    nnn = "true" if layout == "nn" else "false"
    if layout == "nn" and N % 8:
        with pytest.raises(capi.LcError, match="Tensor size mismatch"):
            capi.hgemm_kernel_name(M, N, K, lay, capi.HGEMM_EDGE)
        assert capi.hgemm_kernel_name(M, N, K, lay) == "hgemm_generic_kernel<true>"
        return
    assert capi.hgemm_kernel_name(M, N, K, lay, capi.HGEMM_EDGE) == f"hgemm_edge_kernel<{nnn}>"
    torch.manual_seed(M * 7 + N * 3 + K)
    a = torch.randn(M, K, dtype=torch.half, device="cuda")
    b = torch.randn(K, N, dtype=torch.half, device="cuda")
    bb = host.as_col_major(b) if lay == capi.LAYOUT_TN else b
    pad = 4096                                                       # C inside a larger buffer: nothing outside [0, M * N) is written
    buf = torch.full((M * N + 2 * pad,), float("nan"), dtype=torch.half, device="cuda")
    c = buf[pad:pad + M * N].view(M, N)
    capi.hgemm(a, bb, c, layout=lay, variant=capi.HGEMM_EDGE)
    torch.cuda.synchronize()
    assert torch.isnan(buf[:pad]).all() and torch.isnan(buf[pad + M * N:]).all()
    rows = list(range(M)) if M * N * K <= 1 << 28 else sorted({0, 1, 63, 64, 127, 128, M // 2 + 3, M - 129, M - 2, M - 1})
    truth = oracle.hgemm(a[rows].contiguous(), b.contiguous(), len(rows), N, K, 0, "f32")
    ok, mx, ex = tol.hgemm_close(c[rows].float().cpu().numpy(), truth, K)
    assert ok and torch.isfinite(c).all(), (mx, ex)
    cgen, _ = _run(capi, a, b, lay, capi.HGEMM_GENERIC)
    ulp = torch.clamp(cgen.float().abs(), min=32.0) * 2.0 ** -10
    assert ((c.float() - cgen.float()).abs() <= ulp).all()
    c2, _ = _run(capi, a, b, lay, capi.HGEMM_EDGE)
    assert torch.equal(c2, c)
    if capi.hgemm_kernel_name(M, N, K, lay).startswith("hgemm_edge_kernel"):     # ... and through LC_HGEMM_AUTO where the rule picks it
        c3, _ = _run(capi, a, b, lay, capi.HGEMM_AUTO, 256)
        assert torch.equal(c3, c)


@pytest.mark.parametrize("layout", ["nn", "tn"])
@pytest.mark.parametrize("shape,kind", [((1000, 3000, 512), "mid3"), ((2888, 2880, 544), "mid3"), ((2500, 2504, 96), "mid2"), ((4100, 4104, 320), "w4y"), ((5000, 5008, 288), "w4y"),
                                        ((130, 4232, 96), "mid3"), ((4360, 136, 160), "mid3"), ((100, 4096, 128), "mid3"), ((77, 136, 64), "mid3"),
                                        ((4352, 4104, 96), "w4y"), ((4100, 4352, 352), "w4y")])
def test_ragged_shapes_interior_on_the_tiled_kernels(oracle, layout, shape, kind):
    """Late round 6, LC_HGEMM_RAGGED: M / N that no tile divides with K % 32 == 0 and N % 8 == 0 — more than half a CU's worth of 256 x 256 tiles:
    the interior on hgemm_w4y_kernel (with its ragged last round on the mid-size kernel), the L-shaped border on hgemm_mid_edge_kernel (128 x 128
    tiles of the mid-size kernel that reach beyond M / N: clamped sources, predicated stores; one strip empty when M or N is a multiple of 256);
    otherwise the whole problem on that kernel (three / two ring slots; M < 128 too).  Rows across the seam against the oracle, the whole of C
    against the edge kernel alone and hgemm_generic_kernel (other fp32 orders: one output ulp), bit-identical from run to run, nothing written
    outside C; what LC_HGEMM_AUTO launches follows "hgemm_ragged"."""
    capi = _capi()
    M, N, K = shape
    lay = capi.LAYOUT_NN if layout == "nn" else capi.LAYOUT_TN
    nnn = "true" if layout == "nn" else "false"
    name = capi.hgemm_kernel_name(M, N, K, lay, capi.HGEMM_RAGGED)
    if capi.device_check() == 256:
        if kind == "w4y":
            assert name.startswith(f"hgemm_w4y_kernel<{nnn},") and name.endswith(f" + hgemm_mid_edge_kernel<{nnn},2,2,3>"), name
        else:
            assert name.startswith(f"hgemm_mid_edge_kernel<{nnn},") and name.endswith(",2,2,2>" if kind == "mid2" else ",3>"), name
    torch.manual_seed(M * 7 + N * 3 + K)
    a = torch.randn(M, K, dtype=torch.half, device="cuda")
    b = torch.randn(K, N, dtype=torch.half, device="cuda")
    bb = host.as_col_major(b) if lay == capi.LAYOUT_TN else b
    pad = 4096
    buf = torch.full((M * N + 2 * pad,), float("nan"), dtype=torch.half, device="cuda")
    c = buf[pad:pad + M * N].view(M, N)
    capi.hgemm(a, bb, c, layout=lay, variant=capi.HGEMM_RAGGED, swizzle_stride=host.make_block_swizzle_stride(N, K))
    torch.cuda.synchronize()
    assert torch.isnan(buf[:pad]).all() and torch.isnan(buf[pad + M * N:]).all() and torch.isfinite(c).all()
    mi = M // 256 * 256
    rows = sorted(r for r in {0, 1, 127, 128, 255, 256, M // 2 + 3, mi - 257, mi - 129, mi - 1, mi, mi + 127, mi + 128, M - 2, M - 1} if 0 <= r < M)
    truth = oracle.hgemm(a[rows].contiguous(), b.contiguous(), len(rows), N, K, 0, "f32")
    ok, mx, ex = tol.hgemm_close(c[rows].float().cpu().numpy(), truth, K)
    assert ok, (mx, ex)
    for other in (capi.HGEMM_EDGE, capi.HGEMM_GENERIC):
        co, _ = _run(capi, a, b, lay, other)
        ulp = torch.clamp(co.float().abs(), min=32.0) * 2.0 ** -10
        assert ((c.float() - co.float()).abs() <= ulp).all(), other
    c2, _ = _run(capi, a, b, lay, capi.HGEMM_RAGGED, host.make_block_swizzle_stride(N, K))
    assert torch.equal(c2, c)
    for knob, want in ((1, "hgemm_edge_kernel"), (0, name)):
        capi.tune("hgemm_ragged", knob)
        try:
            assert capi.hgemm_kernel_name(M, N, K, lay).startswith(want)
            c3, _ = _run(capi, a, b, lay, capi.HGEMM_AUTO, host.make_block_swizzle_stride(N, K))
        finally:
            capi.tune("hgemm_ragged", 0)
        if knob == 0:
            assert torch.equal(c3, c)
    if kind != "w4y":   # every tile the layout has ("hgemm_ragged_tile"): one output ulp from the rule's choice, bit-identical from run to run
        for tile in (12, 22, 32) if layout == "nn" else (12, 22, 23, 33):
            capi.tune("hgemm_ragged_tile", tile)
            try:
                tname = capi.hgemm_kernel_name(M, N, K, lay)
                assert tname.startswith(f"hgemm_mid_edge_kernel<{nnn},{tile // 10},{tile % 10},"), (tile, tname)
                ct, _ = _run(capi, a, b, lay, capi.HGEMM_AUTO, 256)
                ct2, _ = _run(capi, a, b, lay, capi.HGEMM_AUTO, 1)
            finally:
                capi.tune("hgemm_ragged_tile", 0)
            assert torch.equal(ct, ct2), tile
            ulp = torch.clamp(c.float().abs(), min=32.0) * 2.0 ** -10
            assert ((ct.float() - c.float()).abs() <= ulp).all(), tile
    # the border launch on the side stream ("hgemm_ragged_fork" 2) / behind the interior (1): the same bits; on a stream of the caller's with
    # the operands still being produced on it (the fork event orders the border behind them, the join event the caller's next kernel behind it)
    for fork in (1, 2):
        capi.tune("hgemm_ragged_fork", fork)
        try:
            s2 = torch.cuda.Stream()
            a2, bb2 = torch.zeros_like(a), torch.zeros_like(bb)
            c4 = torch.full((M, N), float("nan"), dtype=torch.half, device="cuda")
            torch.cuda.synchronize()
            with torch.cuda.stream(s2):
                for _ in range(3):
                    a2.copy_(a)
                    bb2.copy_(bb)
                    capi.hgemm(a2, bb2, c4, layout=lay, variant=capi.HGEMM_RAGGED, swizzle_stride=host.make_block_swizzle_stride(N, K))
                    c5 = c4.clone()                    # (reads C right behind the join)
                    a2.zero_()
                    bb2.zero_()
            torch.cuda.synchronize()
        finally:
            capi.tune("hgemm_ragged_fork", 0)
        assert torch.equal(c5, c), fork
    # under graph capture (no workspace anywhere on this path; never forked)
    cg = torch.full((M, N), float("nan"), dtype=torch.half, device="cuda")
    st = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            capi.hgemm(a, bb, cg, layout=lay, variant=capi.HGEMM_RAGGED, swizzle_stride=host.make_block_swizzle_stride(N, K))
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(cg, c)


@pytest.mark.parametrize("layout", ["nn", "tn"])
def test_ragged_split_k(oracle, layout):
    """Late round 6: split-K of a whole ragged problem (hgemm_mid_edge_sk_kernel + hgemm_mid_reduce_edge_kernel; fp32 partials of whole tiles in the
    stream's workspace): LC_HGEMM_AUTO picks it for one-round grids of 64 / 128 x 128 tiles on at most half the CUs with a long K ("hgemm_mid_splitk",
    the mid-size kernel's own rule); every forced factor against the oracle incl. the K % 64 == 32 half step and K ranges of unequal length, equal to the
    unsplit launch to the rounding of differently grouped fp32 sums, bit-identical from run to run, nothing written outside C; one K range under graph capture."""
    capi = _capi()
    lay = capi.LAYOUT_NN if layout == "nn" else capi.LAYOUT_TN
    nnn = "true" if layout == "nn" else "false"
    if capi.device_check() == 256:
        assert capi.hgemm_kernel_name(100, 4096, 4096, lay) == f"hgemm_mid_edge_sk_kernel<{nnn},1,3> x2"
        assert capi.hgemm_kernel_name(100, 4096, 1024, lay) == f"hgemm_mid_edge_kernel<{nnn},1,2,3>"       # (16 K tiles: never split)
    for (M, N, K) in ((100, 1032, 4096), (77, 136, 8224), (250, 520, 4128)):
        torch.manual_seed(M + N + K)
        a = torch.randn(M, K, dtype=torch.half, device="cuda")
        b = torch.randn(K, N, dtype=torch.half, device="cuda")
        truth = oracle.hgemm(a, b.contiguous(), M, N, K, 0, "f32")
        capi.tune("hgemm_mid_splitk", 1)
        try:
            assert capi.hgemm_kernel_name(M, N, K, lay).startswith(f"hgemm_mid_edge_kernel<{nnn},")
            c1, _ = _run(capi, a, b, lay, capi.HGEMM_AUTO, 1)
        finally:
            capi.tune("hgemm_mid_splitk", 0)
        for tile in (12, 22):
            for ks in (0, 2, 3, 7, 8):
                capi.tune("hgemm_ragged_tile", tile)
                capi.tune("hgemm_mid_splitk", ks)
                try:
                    name = capi.hgemm_kernel_name(M, N, K, lay)
                    bb = host.as_col_major(b) if lay == capi.LAYOUT_TN else b
                    pad = 4096
                    buf = torch.full((M * N + 2 * pad,), float("nan"), dtype=torch.half, device="cuda")
                    c = buf[pad:pad + M * N].view(M, N)
                    capi.hgemm(a, bb, c, layout=lay, variant=capi.HGEMM_AUTO, swizzle_stride=1)
                    torch.cuda.synchronize()
                    c2, _ = _run(capi, a, b, lay, capi.HGEMM_AUTO, 256)
                finally:
                    capi.tune("hgemm_ragged_tile", 0)
                    capi.tune("hgemm_mid_splitk", 0)
                if ks >= 2:
                    assert name == f"hgemm_mid_edge_sk_kernel<{nnn},{tile // 10},3> x{ks}", name
                assert torch.isnan(buf[:pad]).all() and torch.isnan(buf[pad + M * N:]).all()
                assert torch.equal(c, c2), (M, N, K, tile, ks)
                ok, mx, _ = tol.hgemm_close(c.float().cpu().numpy(), truth, K)
                assert ok, (M, N, K, tile, ks, mx)
                ulp = torch.clamp(c1.float().abs(), min=32.0) * 2.0 ** -10
                assert ((c.float() - c1.float()).abs() <= ulp).all(), (M, N, K, tile, ks)
    # under graph capture: one K range, no workspace, the unsplit launch's bits
    M, N, K = 100, 1032, 8192
    a = torch.randn(M, K, dtype=torch.half, device="cuda")
    b = torch.randn(K, N, dtype=torch.half, device="cuda")
    bb = host.as_col_major(b) if lay == capi.LAYOUT_TN else b
    capi.tune("hgemm_mid_splitk", 1)
    try:
        c1, _ = _run(capi, a, b, lay, capi.HGEMM_AUTO, 1)
    finally:
        capi.tune("hgemm_mid_splitk", 0)
    assert " x" in capi.hgemm_kernel_name(M, N, K, lay) or capi.device_check() != 256
    cg = torch.full((M, N), float("nan"), dtype=torch.half, device="cuda")
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        capi.hgemm(a, bb, cg, layout=lay, variant=capi.HGEMM_AUTO)              # (warm-up on the capture stream: split)
        torch.cuda.synchronize()
        cg.fill_(float("nan"))
        with torch.cuda.graph(g, stream=s):
            capi.hgemm(a, bb, cg, layout=lay, variant=capi.HGEMM_AUTO)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(cg, c1)


@pytest.mark.parametrize("layout", ["nn", "tn"])
@pytest.mark.parametrize("shape", [(1000, 3000, 520), (2888, 2880, 264), (512, 1024, 4104), (130, 136, 296)])
def test_k_padding_path(oracle, layout, shape):
    """Late round 6, LC_HGEMM_KPAD: K % 32 != 0 (K % 8 == 0, N % 8 == 0) — A and B copied into the stream's workspace with K zero-padded to a multiple of
    32, the padded problem on LC_HGEMM_AUTO's choice (tiled, ragged; their workspace-free forms).  Zeros add nothing to an fp32 sum: the result equals
    what the same kernel computes on explicitly padded operands BIT FOR BIT; against the oracle and the edge kernel (one output ulp); the source operands
    are not touched; two streams back to back (one buffer per stream); hgemm_edge_kernel under graph capture."""
    capi = _capi()
    M, N, K = shape
    lay = capi.LAYOUT_NN if layout == "nn" else capi.LAYOUT_TN
    Kp = (K + 31) // 32 * 32
    name = capi.hgemm_kernel_name(M, N, K, lay, capi.HGEMM_KPAD)
    assert name == "hgemm_pad_copy_kernel + " + capi.hgemm_kernel_name(M, N, Kp, lay), name
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, dtype=torch.half, device="cuda")
    b = torch.randn(K, N, dtype=torch.half, device="cuda")
    a0, b0 = a.clone(), b.clone()
    c, _ = _run(capi, a, b, lay, capi.HGEMM_KPAD, 256)
    assert torch.equal(a, a0) and torch.equal(b, b0) and torch.isfinite(c).all()
    apad = torch.zeros(M, Kp, dtype=torch.half, device="cuda")
    bpad = torch.zeros(Kp, N, dtype=torch.half, device="cuda")
    apad[:, :K] = a
    bpad[:K] = b
    capi.tune("hgemm_mid_splitk", 1)       # (the padded problem runs its workspace-free form: the operands hold the workspace)
    capi.tune("hgemm_splitk", 1)
    try:
        cp, _ = _run(capi, apad, bpad, lay, capi.HGEMM_AUTO, 256)
    finally:
        capi.tune("hgemm_mid_splitk", 0)
        capi.tune("hgemm_splitk", 0)
    if " x" not in capi.hgemm_kernel_name(M, N, Kp, lay):
        assert torch.equal(c, cp)
    else:   # (the padded problem would split K through the workspace: inside LC_HGEMM_KPAD it runs the unsplit launch of the same tile, the knob = 1 launch above may be another kernel)
        ulp = torch.clamp(cp.float().abs(), min=32.0) * 2.0 ** -10
        assert ((c.float() - cp.float()).abs() <= ulp).all()
    truth = oracle.hgemm(a, b.contiguous(), M, N, K, 0, "f32") if M * N * K <= 1 << 31 else None
    if truth is not None:
        ok, mx, ex = tol.hgemm_close(c.float().cpu().numpy(), truth, K)
        assert ok, (mx, ex)
    ce, _ = _run(capi, a, b, lay, capi.HGEMM_EDGE)
    ulp = torch.clamp(ce.float().abs(), min=32.0) * 2.0 ** -10
    assert ((c.float() - ce.float()).abs() <= ulp).all()
    # two streams, several launches each, operands rewritten between launches
    outs = []
    bb = host.as_col_major(b) if lay == capi.LAYOUT_TN else b
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    for it in range(3):
        for st in streams:
            with torch.cuda.stream(st):
                ci = torch.full((M, N), float("nan"), dtype=torch.half, device="cuda")
                capi.hgemm(a, bb, ci, layout=lay, variant=capi.HGEMM_KPAD, swizzle_stride=256)
                outs.append(ci)
    torch.cuda.synchronize()
    assert all(torch.equal(o, c) for o in outs)
    # LC_HGEMM_AUTO follows "hgemm_kpad" (2 = wherever legal); under graph capture: no workspace -> the edge kernel
    capi.tune("hgemm_kpad", 2)
    try:
        assert capi.hgemm_kernel_name(M, N, K, lay) == name
        c2, _ = _run(capi, a, b, lay, capi.HGEMM_AUTO, 256)
        cg = torch.full((M, N), float("nan"), dtype=torch.half, device="cuda")
        s = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            capi.hgemm(a, bb, cg, layout=lay, variant=capi.HGEMM_AUTO)
            torch.cuda.synchronize()
            cg.fill_(float("nan"))
            with torch.cuda.graph(g, stream=s):
                capi.hgemm(a, bb, cg, layout=lay, variant=capi.HGEMM_AUTO)
        g.replay()
        torch.cuda.synchronize()
    finally:
        capi.tune("hgemm_kpad", 0)
    assert torch.equal(c2, c) and torch.equal(cg, ce)


@pytest.mark.parametrize("variant", ["mfma256", "pingpong2", "w4b", "w4c", "w4x", "w4y", "generic"])
def test_identity_times_asymmetric_b_detects_transposes(variant):
    capi = _capi()
    n = 512
    a = torch.eye(n, dtype=torch.half, device="cuda")
    b = (torch.arange(n * n, device="cuda").reshape(n, n) % 1021).half() / 4   # asymmetric, exact
    for lay in (capi.LAYOUT_NN, capi.LAYOUT_TN):
        c, _ = _run(capi, a, b, lay, VARIANTS[variant])
        assert torch.equal(c, b)
        c2, _ = _run(capi, b, a, lay, VARIANTS[variant])
        assert torch.equal(c2, b)


def test_golden_fixtures_and_reference_distribution(oracle, golden):
    """Inputs of the reference's torch baseline (tests/golden) and its C++ harness distribution
    (uniform {-1.00..0.99} step 0.01, kernels/hgemm/utils/utils.h:238)."""
    capi = _capi()
    g = golden["hgemm"]
    for i in range(4):
        a = torch.from_numpy(g[f"a{i}"].view(np.float16)).cuda()
        b = torch.from_numpy(g[f"b{i}"].view(np.float16)).cuda()
        bcol = torch.from_numpy(g[f"bcol{i}"].view(np.float16)).cuda()
        M, K = a.shape
        N = b.shape[1]
        for lay, bb in ((capi.LAYOUT_NN, b), (capi.LAYOUT_TN, bcol)):
            c = torch.empty(M, N, dtype=torch.half, device="cuda")
            capi.hgemm(a, bb, c, layout=lay, variant=capi.HGEMM_AUTO)
            torch.cuda.synchronize()
            ok, mx, _ = tol.hgemm_close(c.float().cpu().numpy(), g[f"c64_{i}"], K)
            assert ok, mx
    gen = torch.Generator().manual_seed(11)
    a = ((torch.randint(0, 200, (512, 512), generator=gen) - 100).float() * 0.01).half().cuda()
    b = ((torch.randint(0, 200, (512, 512), generator=gen) - 100).float() * 0.01).half().cuda()
    c, _ = _run(capi, a, b, capi.LAYOUT_NN, capi.HGEMM_AUTO)
    _check(oracle, capi, a, b, c, capi.LAYOUT_NN, amp=0.58)


def test_every_reference_entry_name_computes_the_gemm(oracle):
    capi = _capi()
    M = N = K = 256
    torch.manual_seed(5)
    a = torch.randn(M, K, dtype=torch.half, device="cuda")
    b = torch.randn(K, N, dtype=torch.half, device="cuda")
    bcol = host.as_col_major(b)
    truth = oracle.hgemm(a, b, M, N, K, 0, "f32")
    capi.hgemm_call("init_cublas_handle", a, b, a)
    try:
        for name, lay, nargs in capi.hgemm_entries():
            if nargs == 0:
                continue
            c = torch.zeros(M, N, dtype=torch.half, device="cuda")
            bb = bcol if lay == capi.LAYOUT_TN else b
            stride = host.make_block_swizzle_stride(N, K)
            capi.hgemm_call(name, a, bb, c, stages=3, swizzle=True, swizzle_stride=stride)
            torch.cuda.synchronize()
            ok, mx, _ = tol.hgemm_close(c.float().cpu().numpy(), truth, K)
            assert ok, (name, mx)
        # reference-legal but not 256-tileable shape (multiples of 128, K of 32) still resolves
        a2 = torch.randn(384, 96, dtype=torch.half, device="cuda")
        b2 = torch.randn(96, 128, dtype=torch.half, device="cuda")
        c2 = torch.zeros(384, 128, dtype=torch.half, device="cuda")
        capi.hgemm_call("hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem", a2, b2, c2, 2, False, 1)
        torch.cuda.synchronize()
        ok, mx, _ = tol.hgemm_close(c2.float().cpu().numpy(), oracle.hgemm(a2, b2, 384, 128, 96, 0, "f32"), 96)
        assert ok, mx
    finally:
        capi.hgemm_call("destroy_cublas_handle", a, b, a)


def test_cross_check_entry_names_run_the_auto_kernel_on_small_grids():
    """Round 6: the entry names that map to the 256-tile cross-check kernels / the 128-tile kernel keep them on large grids (the reference
    bench's rows stay distinct there) but run what LC_HGEMM_AUTO runs where a 256 x 256 tile serves nobody — 2048^3 is 64 such tiles on 256
    CUs: 330 TFLOP/s against 818 in the reference's own unmodified sweep (profiles/r6Z_f1_hgemm_default_sweep.log) — so a drop-in caller
    of ANY name gets the mid-size kernel's bits and speed there."""
    capi = _capi()
    n = 2048
    torch.manual_seed(2048)
    a = torch.randn(n, n, dtype=torch.half, device="cuda")
    b = torch.randn(n, n, dtype=torch.half, device="cuda")
    bcol = host.as_col_major(b)
    ref = {}
    for lay, bb in ((capi.LAYOUT_NN, b), (capi.LAYOUT_TN, bcol)):
        c = torch.zeros(n, n, dtype=torch.half, device="cuda")
        capi.hgemm(a, bb, c, layout=lay, variant=capi.HGEMM_AUTO, swizzle_stride=1024)
        ref[lay] = c
    torch.cuda.synchronize()
    for name in ("hgemm_mma_m16n8k16_mma2x4_warp4x4_stages_dsmem_tn", "hgemm_mma_m16n8k16_mma2x4_warp4x4_stages_dsmem", "hgemm_mma_m16n8k16_mma2x4_warp4x4",
                 "hgemm_wmma_m16n16k16_mma4x2_warp2x4_stages_dsmem", "hgemm_wmma_m16n16k16_mma4x4_warp4x4_stages_dsmem", "hgemm_wmma_m16n16k16_mma4x2_warp2x4"):
        lay = dict((nm, l) for nm, l, _ in capi.hgemm_entries())[name]
        c = torch.full((n, n), float("nan"), dtype=torch.half, device="cuda")
        capi.hgemm_call(name, a, bcol if lay == capi.LAYOUT_TN else b, c, 2, True, 1024)
        torch.cuda.synchronize()
        assert torch.equal(c, ref[lay]), name


@pytest.mark.parametrize("rung", list(range(20, 31)))
def test_vector_alu_ladder_rungs(oracle, rung):
    """f2: the reference's CUDA-core ladder (naive/hgemm.cu) as real vector-ALU kernels (hgemm_valu.hip): every rung against
    the oracle on a shape all of them tile (incl. the 256-row t_16x8 rung), the dispatcher reporting the rung's own kernel,
    and the documented fall-backs (TN, shapes a rung does not tile -> the edge kernel; the naive rung takes ANY shape)."""
    capi = _capi()
    M, N, K = 512, 384, 320
    torch.manual_seed(rung)
    a = torch.randn(M, K, dtype=torch.half, device="cuda")
    b = torch.randn(K, N, dtype=torch.half, device="cuda")
    c = torch.zeros(M, N, dtype=torch.half, device="cuda")
    name = capi.hgemm_kernel_name(M, N, K, capi.LAYOUT_NN, rung)
    assert name.startswith("hgemm_valu_"), name
    capi.hgemm(a, b, c, layout=capi.LAYOUT_NN, variant=rung)
    torch.cuda.synchronize()
    ok, mx, _ = tol.hgemm_close(c.float().cpu().numpy(), oracle.hgemm(a, b, M, N, K, 0, "f32"), K)
    assert ok, (rung, mx)
    assert capi.hgemm_kernel_name(M, N, K, capi.LAYOUT_TN, rung) == "hgemm_generic_kernel<false>"
    M2, N2, K2 = 100, 72, 50          # tiles nothing
    a2 = torch.randn(M2, K2, dtype=torch.half, device="cuda")
    b2 = torch.randn(K2, N2, dtype=torch.half, device="cuda")
    c2 = torch.zeros(M2, N2, dtype=torch.half, device="cuda")
    want = "hgemm_valu_naive_kernel" if rung == 20 else "hgemm_generic_kernel<true>"
    assert capi.hgemm_kernel_name(M2, N2, K2, capi.LAYOUT_NN, rung) == want
    capi.hgemm(a2, b2, c2, layout=capi.LAYOUT_NN, variant=rung)
    torch.cuda.synchronize()
    ok, mx, _ = tol.hgemm_close(c2.float().cpu().numpy(), oracle.hgemm(a2, b2, M2, N2, K2, 0, "f32"), K2)
    assert ok, (rung, mx)


def test_torch_extension_module_drop_in(oracle):
    """The reference's call convention end to end: import toy_hgemm; f(a, b, c, stages, swizzle, stride)."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "leetcuda_amd"))
    import toy_hgemm
    M = N = K = 512
    torch.manual_seed(9)
    a = torch.randn(M, K, dtype=torch.half, device="cuda")
    b = torch.randn(K, N, dtype=torch.half, device="cuda")
    c = torch.zeros(M, N, dtype=torch.half, device="cuda")
    truth = oracle.hgemm(a, b, M, N, K, 0, "f32")
    toy_hgemm.hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem(a, b, c, 2, True, 256)
    torch.cuda.synchronize()
    assert tol.hgemm_close(c.float().cpu().numpy(), truth, K)[0]
    c.zero_()
    toy_hgemm.hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_tn_swizzle_x4(a, host.as_col_major(b), c, 2, True, 256)
    torch.cuda.synchronize()
    assert tol.hgemm_close(c.float().cpu().numpy(), truth, K)[0]
    toy_hgemm.init_cublas_handle()
    c.zero_()
    toy_hgemm.hgemm_cublas_tensor_op_nn(a, b, c)
    toy_hgemm.destroy_cublas_handle()
    torch.cuda.synchronize()
    assert tol.hgemm_close(c.float().cpu().numpy(), truth, K)[0]
    with pytest.raises(RuntimeError, match="Tensor size mismatch!"):
        toy_hgemm.hgemm_naive_f16(a, b[:256], c)
    # boundary hardening (round-4 verdict, structure #11 / #12): views are refused, the launch follows the tensors' device and ITS
    # current stream (here: a side stream of device 0 entered without touching the current device)
    with pytest.raises(RuntimeError, match="must be contiguous"):
        toy_hgemm.hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_tn_swizzle_x4(a, b.t(), c, 2, True, 256)
    side = torch.cuda.Stream(device=0)
    c.zero_()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        toy_hgemm.hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem(a, b, c, 2, True, 256)
    side.synchronize()               # ONLY the side stream: the result must be there
    assert tol.hgemm_close(c.float().cpu().numpy(), truth, K)[0]


@pytest.mark.parametrize("layout", ["nn", "tn"])
def test_full_size_config2_properties(oracle, layout):
    """BASELINE config 2: M=N=K=8192, randn fp16 exactly as hgemm.py:444-446."""
    capi = _capi()
    n = 8192
    torch.manual_seed(0)
    a = torch.randn(n, n, dtype=torch.half, device="cuda")
    b = torch.randn(n, n, dtype=torch.half, device="cuda")
    lay = capi.LAYOUT_NN if layout == "nn" else capi.LAYOUT_TN
    stride = host.make_block_swizzle_stride(n, n)
    outs = {}
    for name, var in VARIANTS.items():
        if name == "generic":
            continue
        outs[name], _ = _run(capi, a, b, lay, var, stride)
    # (1) the independently scheduled kernels agree to fp16 rounding (accumulation order differs -> a rounding
    #     flip = 1 ulp): 8-wave one-barrier, 8-wave ping-pong, and the 4-wave ring kernels (glds / buffer / spread DMA)
    ref = outs["pingpong2"].float()
    ulp = torch.clamp(ref.abs(), min=64.0) * 2.0 ** -10
    for name in ("mfma256", "w4b", "w4c", "w4x", "w4y"):
        assert ((outs[name].float() - ref).abs() <= ulp).all(), name
    assert torch.equal(outs["w4b"], outs["w4c"])   # same MFMA order, glds vs buffer DMA
    if layout == "tn":
        # 16x16x32 kernels, compiler-scheduled vs hand-ordered stream: same bits when both walk K from tile 0 (the default
        # hgemm_w4y launch staggers the walk per XCD: same products, another fp32 summation order)
        capi.tune("hgemm_stagger", 1 << 27)
        try:
            plain, _ = _run(capi, a, b, lay, VARIANTS["w4y"], stride)
        finally:
            capi.tune("hgemm_stagger", 0)
        assert torch.equal(outs["w4x"], plain)
        assert not torch.equal(outs["w4y"], plain) and ((outs["w4y"].float() - plain.float()).abs() <= ulp).all()
        for sched in (0, 1):                            # the other generated schedules of the loop body (2 = the default): same bits
            capi.tune("w4y_sched", sched)
            try:
                alt, _ = _run(capi, a, b, lay, VARIANTS["w4y"], stride)
            finally:
                capi.tune("w4y_sched", 2)
            assert torch.equal(alt, outs["w4y"]), sched
    c = outs["w4c"]
    # (2) C·x == A·(B·x) in fp64 on the host, x random: any wrong tile shifts thousands of entries
    rng = np.random.default_rng(0)
    x = rng.standard_normal(n)
    an, bn = a.cpu().numpy().astype(np.float32), b.cpu().numpy().astype(np.float32)
    want = an.astype(np.float64) @ (bn.astype(np.float64) @ x)
    got = c.cpu().numpy().astype(np.float64) @ x
    rel = np.abs(got - want).max() / np.abs(want).max()
    assert rel < 2e-3, rel
    # (3) sampled rows against the exact oracle (full K)
    rows = [0, 255, 256, 4097, 8191]
    truth = oracle.hgemm(a[rows].contiguous(), b, len(rows), n, n, 0, "f32")
    ok, mx, _ = tol.hgemm_close(c[rows].float().cpu().numpy(), truth, n)
    assert ok, mx
    # (4) vendor comparator (hipBLASLt, fp32 compute) agrees
    capi.vendor_init()
    try:
        cv = torch.empty_like(c)
        capi.hgemm_vendor(a, host.as_col_major(b) if lay == capi.LAYOUT_TN else b, cv, lay)
        torch.cuda.synchronize()
        ok, mx, _ = tol.hgemm_close(cv[rows].float().cpu().numpy(), truth, n)
        assert ok, mx
    finally:
        capi.vendor_destroy()


@pytest.mark.parametrize("layout", ["nn", "tn"])
def test_k_loop_stagger_walks_every_k_tile_once(oracle, layout):
    """lc_tune_set "hgemm_stagger" (hgemm_w4y.hip): a workgroup starts its K walk at tile ((index & mask) * step) mod KT and wraps —
    every product is still summed exactly once, only the fp32 accumulation order changes.  The default (0: by XCD, an eighth of K
    apart), XCD / row / column / mixed indices, start tiles
    beyond KT (reduced mod KT), KT = 1 / 2 / 3 (the clamped prefetch of the last iterations wraps too), against the unstaggered
    walk (fp16 rounding) and the exact oracle; a K-dependent input (column k of A scaled by a ramp) makes a skipped or doubled
    tile visible in every element."""
    capi = _capi()
    lay = capi.LAYOUT_NN if layout == "nn" else capi.LAYOUT_TN
    for M, N, K in ((512, 768, 64), (512, 512, 128), (768, 512, 192), (1024, 1024, 1024), (2048, 2304, 4096)):
        torch.manual_seed(K + M)
        ramp = 1.0 + torch.arange(K, device="cuda").float() / K
        a = (torch.randn(M, K, device="cuda") * ramp[None, :]).half()
        b = torch.randn(K, N, dtype=torch.half, device="cuda")
        capi.tune("hgemm_stagger", 1 << 27)          # reference: the plain walk from tile 0
        try:
            ref, _ = _run(capi, a, b, lay, VARIANTS["w4y"], 2048)
        finally:
            capi.tune("hgemm_stagger", 0)
        rows = [0, M // 2 + 1, M - 1]
        truth = oracle.hgemm(a[rows].contiguous(), b, len(rows), N, K, 0, "f32")
        ulp = torch.clamp(ref.float().abs(), min=64.0) * 2.0 ** -10
        for knob in (0, 1 | 16 << 12 | 7 << 20, 1 << 4 | 2 << 12 | 31 << 20, 1 << 8 | 2 << 12 | 31 << 20, 1 << 4 | 3 << 8 | 3 << 12 | 31 << 20,
                     15 | 15 << 4 | 15 << 8 | 255 << 12 | 127 << 20):     # (mask is 7 bits: bit 27 is the "off" value)
            capi.tune("hgemm_stagger", knob)
            try:
                got, _ = _run(capi, a, b, lay, VARIANTS["w4y"], 2048)
            finally:
                capi.tune("hgemm_stagger", 0)
            assert torch.isfinite(got).all()
            assert ((got.float() - ref.float()).abs() <= ulp).all(), (M, N, K, knob)
            ok, mx, _ = tol.hgemm_close(got[rows].float().cpu().numpy(), truth, K)
            assert ok, (M, N, K, knob, mx)


@pytest.mark.parametrize("layout", ["nn", "tn"])
def test_persistent_workgroup_walk_computes_the_same_bits(layout):
    """lc_tune_set "hgemm_persist" = 1: hgemm_w4y_kernel as one persistent workgroup per CU that walks the C tiles w, w + G, ... (taken
    when the tile count is a multiple of the CU count and larger).  The virtual block id of a tile — hence its raster position and
    its K-loop stagger — is what the one-tile launch gives it: outputs must be IDENTICAL, with both block -> tile maps."""
    capi = _capi()
    lay = capi.LAYOUT_NN if layout == "nn" else capi.LAYOUT_TN
    for M, N, K in ((8192, 4096, 256), (4096, 8192, 448), (8192, 8192, 128), (8192, 8192, 64)):      # K = 64: ONE K tile per C tile — the cross-tile prefetch is the whole K loop's supply
        torch.manual_seed(M + K)
        a = torch.randn(M, K, dtype=torch.half, device="cuda")
        b = torch.randn(K, N, dtype=torch.half, device="cuda")
        for raster in (1, 2):
            outs = []
            for persist in (0, 1):
                capi.tune("hgemm_persist", persist)
                capi.tune("hgemm_raster", raster)
                try:
                    c, _ = _run(capi, a, b, lay, VARIANTS["w4y"], 2048)
                finally:
                    capi.tune("hgemm_persist", 1)      # the library default (round 3 restored 0 here: every later test ran the non-default launch)
                    capi.tune("hgemm_raster", 0)
                outs.append(c)
            assert torch.isfinite(outs[1]).all() and torch.equal(outs[0], outs[1]), (M, N, K, raster)


def test_auto_large_nn_b_uses_64bit_dma_addresses():
    """B of 2 GiB (NN): the buffer-descriptor DMA of the AUTO kernel (32-bit offsets) must hand over to the 64-bit
    global form; the result has to agree with the independently scheduled 8-wave kernel on every element."""
    capi = _capi()
    M, N, K = 512, 32768, 32768
    torch.manual_seed(5)
    a = torch.randn(M, K, dtype=torch.half, device="cuda")
    b = torch.randn(K, N, dtype=torch.half, device="cuda")
    c1, _ = _run(capi, a, b, capi.LAYOUT_NN, VARIANTS["w4c"], 2048)      # falls back to w4b internally
    c2, _ = _run(capi, a, b, capi.LAYOUT_NN, VARIANTS["pingpong2"], 2048)
    ulp = torch.clamp(c2.float().abs(), min=64.0) * 2.0 ** -10
    assert ((c1.float() - c2.float()).abs() <= ulp).all()
    # and the last K tile really contributed: perturbing the last row of B changes C
    b[-1].add_(1.0)
    c3, _ = _run(capi, a, b, capi.LAYOUT_NN, 0, 2048)
    assert (c3.float() - c1.float()).abs().max().item() > 0.1


@pytest.mark.parametrize("layout", ["nn", "tn"])
def test_xcd_super_block_raster_computes_the_same_bits(layout):
    """lc_tune_set "hgemm_raster" = 1 only changes WHICH workgroup computes which C tile (hgemm_mfma256.hip raster_xcd16):
    outputs must equal the block-swizzle raster's bit for bit on full 16 x 16 tile grids, ragged grids (tile rows / columns not
    a multiple of 16, < 256 trailing blocks) and the 128-tile kernel's grid."""
    from leetcuda_amd import capi, host
    capi.load()
    lay = capi.LAYOUT_NN if layout == "nn" else capi.LAYOUT_TN
    for M, N, K in ((4096, 4096, 128), (3072, 3328, 128), (256 * 17, 256 * 9, 64), (256 * 33, 256 * 18, 64), (1024, 1280, 256)):
        torch.manual_seed(M + N)
        a = torch.randn(M, K, dtype=torch.half, device="cuda")
        b = torch.randn(K, N, dtype=torch.half, device="cuda")
        bb = host.as_col_major(b) if lay == capi.LAYOUT_TN else b
        outs = []
        for knob in (1, 2):
            capi.tune("hgemm_raster", knob)
            capi.tune("hgemm_stagger", 1 << 27)     # (the default K walk starts per XCD: another map = another summation order per tile)
            try:
                c = torch.full((M, N), float("nan"), dtype=torch.half, device="cuda")
                capi.hgemm(a, bb, c, layout=lay, variant=capi.HGEMM_AUTO, swizzle_stride=1024)
                torch.cuda.synchronize()
            finally:
                capi.tune("hgemm_raster", 0)
                capi.tune("hgemm_stagger", 0)
            outs.append(c)
        assert torch.isfinite(outs[1]).all() and torch.equal(outs[0], outs[1]), (M, N, K)


@pytest.mark.parametrize("layout", ["nn", "tn"])
@pytest.mark.parametrize("raster", [1, 2])
def test_ragged_last_wave_goes_to_the_128_tile_kernel(oracle, layout, raster):
    """lc_tune_set "hgemm_tail" = 1 (default): when the 256-tile grid's last wave holds at most 128 tiles, hgemm_w4y_kernel runs the
    full waves and 128 x 128 blocks the four quadrants of every remaining tile (lc_abi.hip launch_mfma256) — since round 6 on
    hgemm_mid_kernel (three ring slots when the blocks fit one round, two beyond; no workspace; 64 x 128 eighths instead of quadrants while
    THEY fit one round, "hgemm_tail_tile"), "hgemm_tail" = 2 keeps round 5's
    hgemm_mfma128_kernel + split-K form.  Every element is written exactly once, all three agree with the one-launch result to fp16
    rounding of differently ordered fp32 sums — the two quadrant kernels with each other bit for bit when the 128-tile kernel does not
    split K — and sampled rows match the oracle, with both block -> tile maps (the remainder ids differ between them)."""
    from leetcuda_amd import capi, host
    capi.load()
    lay = capi.LAYOUT_NN if layout == "nn" else capi.LAYOUT_TN
    for M, N, K in ((256 * 18, 256 * 18, 256), (256 * 17, 256 * 19, 192), (256 * 9, 256 * 29, 128), (256 * 17, 256 * 17, 96)):   # T % 256 = 68, 67, 5, 33
        assert 0 < (M // 256) * (N // 256) % 256 <= 128
        torch.manual_seed(M + N + K)
        a = torch.randn(M, K, dtype=torch.half, device="cuda")
        b = torch.randn(K, N, dtype=torch.half, device="cuda")
        bb = host.as_col_major(b) if lay == capi.LAYOUT_TN else b
        outs = []
        capi.tune("hgemm_raster", raster)
        try:
            for tail, splitk, tile in ((0, 0, 0), (1, 0, 2), (2, 1, 0), (1, 0, 1), (1, 0, 0)):
                capi.tune("hgemm_tail", tail)
                capi.tune("hgemm_tail_tile", tile)     # (quadrants, 64 x 128 eighths, the rule's choice: same sums in the same order)
                capi.tune("hgemm_splitk", splitk)      # (2: round 5's kernel without its split-K and on four waves, so that it sums k in the
                capi.tune("hgemm_128w", 1 if tail == 2 else 0)    #  mid-size kernel's order)
                c = torch.full((M, N), float("nan"), dtype=torch.half, device="cuda")
                capi.hgemm(a, bb, c, layout=lay, variant=capi.HGEMM_MFMA256W4Y, swizzle_stride=2048)
                torch.cuda.synchronize()
                outs.append(c)
        finally:
            capi.tune("hgemm_tail", 1)
            capi.tune("hgemm_tail_tile", 0)
            capi.tune("hgemm_splitk", 0)
            capi.tune("hgemm_128w", 0)
            capi.tune("hgemm_raster", 0)
        assert torch.isfinite(outs[1]).all() and torch.isfinite(outs[2]).all()
        assert torch.equal(outs[1], outs[2])          # mid-size kernel == 128-tile kernel on the quadrants, bit for bit
        assert torch.equal(outs[1], outs[3]) and torch.equal(outs[1], outs[4])   # ... and its 64 x 128 eighths / the rule's choice too
        d = (outs[0].float() - outs[1].float()).abs()
        # (the two kernels' fp32 accumulation orders differ — 16x16x32 vs 32x32x16 chains — yet on these shapes they round to the same
        #  fp16 almost everywhere, often everywhere: the bound below is what is asserted, not a difference)
        assert d.max().item() <= 0.0625 and (d > 0).float().mean().item() < 0.2, (d.max().item(), (d > 0).float().mean().item())
        rows = [0, 255, 256, M // 2 + 3, M - 1]
        truth = oracle.hgemm(a[rows].contiguous(), b, len(rows), N, K, 0, "f32")
        ok, mx, _ = tol.hgemm_close(outs[1][rows].float().cpu().numpy(), truth, K)
        assert ok, mx


def test_concurrent_host_threads_on_two_streams():
    """Round-3 advisor finding: the ragged-tail block count used to travel through a file-scope global, so two host threads launching
    different shapes could pick up each other's truncated grid and leave part of C unwritten.  It is a launcher argument now: three
    threads (ctypes releases the GIL inside the call), each with its own stream and shape — 504 tiles (last wave 248: one launch, one
    tile per workgroup), 512 tiles (persistent walk with the cross-tile prefetch), 324 tiles (last wave 68 <= 128: the tail split,
    256 tiles on the generated-loop kernel + 272 quadrants on the 128-tile kernel) — launch 60 GEMMs each at the same time; every
    checked output must equal the single-threaded result bit for bit, with no unwritten (NaN-prefilled) element."""
    import threading
    capi = _capi()
    shapes = [(6144, 5376, 256), (8192, 4096, 320), (4608, 4608, 256)]
    data, want = [], []
    for M, N, K in shapes:
        torch.manual_seed(M + N)
        a = torch.randn(M, K, dtype=torch.half, device="cuda")
        b = host.as_col_major(torch.randn(K, N, dtype=torch.half, device="cuda"))
        c = torch.full((M, N), float("nan"), dtype=torch.half, device="cuda")
        capi.hgemm(a, b, c, layout=capi.LAYOUT_TN, variant=capi.HGEMM_AUTO, swizzle_stride=2048)
        torch.cuda.synchronize()
        assert torch.isfinite(c).all()
        data.append((a, b))
        want.append(c)
    errs = []

    def worker(idx):
        try:
            st = torch.cuda.Stream()
            a, b = data[idx]
            M, N = want[idx].shape
            with torch.cuda.stream(st):
                for it in range(60):
                    c = torch.full((M, N), float("nan"), dtype=torch.half, device="cuda")
                    capi.hgemm(a, b, c, layout=capi.LAYOUT_TN, variant=capi.HGEMM_AUTO, swizzle_stride=2048)
                    if it % 10 == 9:
                        st.synchronize()
                        if not torch.equal(c, want[idx]):
                            errs.append((idx, it))
            st.synchronize()
        except Exception as e:      # noqa: BLE001
            errs.append((idx, repr(e)))

    ths = [threading.Thread(target=worker, args=(i,)) for i in range(len(shapes))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    torch.cuda.synchronize()
    assert not errs, errs


def test_random_shapes_through_the_auto_dispatch(oracle):
    """Thirty random (M, N, K, layout) with M, N multiples of 64 and K of 32 — whatever LC_HGEMM_AUTO picks (eight-wave 128-tile kernel,
    the mid-size kernel in any tile / ring depth, the 256-tile kernel with border strips and half K-steps, the edge kernel for the
    64-multiples no 128-tile divides) must match the oracle; the kernel families seen are collected so that a change of the rules shows up
    here (round 6: the mid-size kernel must be among them)."""
    capi = _capi()
    rng = np.random.default_rng(6)
    seen = set()
    for i in range(30):
        M = int(rng.integers(2, 41)) * 64
        N = int(rng.integers(2, 41)) * 64
        K = int(rng.integers(2, 66)) * 32
        if i % 3 == 0:                       # a third of them in the mid-size kernel's home range
            M, N = int(rng.integers(8, 23)) * 128, int(rng.integers(8, 23)) * 128
        lay = capi.LAYOUT_NN if rng.random() < 0.5 else capi.LAYOUT_TN
        torch.manual_seed(M + 3 * N + 7 * K)
        a = torch.randn(M, K, dtype=torch.half, device="cuda")
        b = torch.randn(K, N, dtype=torch.half, device="cuda")
        name = capi.hgemm_kernel_name(M, N, K, lay)
        seen.add(name.split("<")[0])
        c, _ = _run(capi, a, b, lay, capi.HGEMM_AUTO, host.make_block_swizzle_stride(N, K))
        truth = oracle.hgemm(a, b, M, N, K, 0, "f32")
        ok, mx, ex = tol.hgemm_close(c.float().cpu().numpy(), truth, K)
        assert ok, (M, N, K, lay, name, mx, ex)
    assert {"hgemm_mid_kernel", "hgemm_mid_edge_kernel"} <= seen and ({"hgemm_mfma128_kernel", "hgemm_w4y_kernel"} & seen), seen
