"""GPU parity tests of the FlashAttention-2 forward path through the C-ABI against the CPU oracle."""
import numpy as np
import pytest
import torch

from tests import tol

pytestmark = pytest.mark.gpu


def _capi():
    from leetcuda_amd import capi
    capi.load()
    return capi


def _check(oracle, q, k, v, o, vt=False, max_abs=tol.ATTN_MAX_ABS):
    B, H, N, D = q.shape
    truth = oracle.attn(q, k, v, B, H, N, D, vt=vt, mode="f32")
    out = o.float().cpu().numpy()
    assert np.isfinite(out).all()
    diff = np.abs(out - truth)
    # the reference's own --check: torch.allclose(ref, out, atol=1e-2) (flash_attn_mma.py:489)
    assert np.allclose(out, truth, atol=tol.ATTN_ATOL, rtol=1e-5), diff.max()
    assert diff.max() < max_abs, (diff.max(), diff.mean())
    return diff.max(), diff.mean()


@pytest.mark.parametrize("D", [32, 64, 96, 128])
@pytest.mark.parametrize("N", [64, 128, 256, 512])
def test_attn_vs_oracle(oracle, D, N):
    capi = _capi()
    B, H = 2, 3
    torch.manual_seed(D * 1000 + N)
    q = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    k = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    v = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    o = torch.full_like(q, float("nan"))
    capi.attn_fwd(q, k, v, o)
    torch.cuda.synchronize()
    _check(oracle, q, k, v, o)
    # V handed over transposed ([B,H,D,N], the *_swizzle_qkv convention) gives the same result
    tv = v.transpose(-2, -1).contiguous()
    o2 = torch.full_like(q, float("nan"))
    capi.attn_fwd(q, k, tv, o2, v_transposed=True)
    torch.cuda.synchronize()
    _check(oracle, q, k, tv, o2, vt=True)
    assert (o.float() - o2.float()).abs().max().item() < 1e-3


@pytest.mark.parametrize("nw", [512, 513, 515, 517, 514, 8, 4, 2])
@pytest.mark.parametrize("D", [128, 64, 96, 32])
def test_workgroup_shapes_agree(oracle, nw, D):
    """The same problem through the merged-phase 4-wave kernel (attn_w4u.hip: 513 / 515 / 517 = one block per workgroup /
    persistent static walk / persistent dynamic queue; 512 = round 2's name for 513) and the 8-, 4-, 2-wave lock-step kernels
    (lc_tune_set "attn_nw"; 514 = the same design with each phase as one generated asm statement, attn_w4i.hip); D = 96 / 32 run
    the generated kernel or the lock-step kernel."""
    capi = _capi()
    B, H, N = 1, 3, 768
    torch.manual_seed(77 + D)
    q = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    k = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    v = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    o = torch.zeros_like(q)
    capi.tune("attn_nw", nw)
    try:
        capi.attn_fwd(q, k, v, o)
        torch.cuda.synchronize()
    finally:
        capi.tune("attn_nw", 0)
    _check(oracle, q, k, v, o)


@pytest.mark.parametrize("D,N", [(256, 128), (256, 192), (256, 256), (256, 512), (256, 384), (512, 256), (512, 128), (512, 64), (1024, 128), (1024, 64),
                                 (1024, 192)])   # (smallest launches: (256,256) = 8 ring periods on attn_bigd7, (512,128) on attn_bigd6, (1024,64) on attn_bigd4; (256,384): attn_bigd2)
def test_large_head_dims_tiling_qkv(oracle, D, N):
    """D = 256 / 512 / 1024: the fine-grained Q,K,V d-slice tiling (reference: flash_attn_mma_tiling_qkv.cu,
    dispatcher cases 256, 512, 1024), through the FFPA-ancestor entry names."""
    capi = _capi()
    B, H = 1, 2
    torch.manual_seed(D + N)
    q = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    k = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    v = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    for name in ("flash_attn_mma_stages_split_q_tiling_qkv", "flash_attn_mma_stages_split_q_tiling_qkv_acc_f32"):
        o = torch.full_like(q, float("nan"))
        capi.attn_call(name, q, k, v, o, 2)
        torch.cuda.synchronize()
        _check(oracle, q, k, v, o)
    if D <= 256:   # V handed over as [B,H,D,N] (tiling_qk_swizzle_qkv: d <= 256)
        tv = v.transpose(-2, -1).contiguous()
        o = torch.full_like(q, float("nan"))
        capi.attn_call("flash_attn_mma_stages_split_q_tiling_qk_swizzle_qkv", q, k, tv, o, 2)
        torch.cuda.synchronize()
        _check(oracle, q, k, tv, o, vt=True)
        if N % 256 == 0:   # auto hands a grid this small to attn_bigd2 (128-row workgroups): attn_bigd7 on it, both V layouts ("attn_d512" = 4)
            capi.tune("attn_d512", 4)
            try:
                for args, vt in (((q, k, v), False), ((q, k, tv), True)):
                    o = torch.full_like(q, float("nan"))
                    capi.attn_call("flash_attn_mma_stages_split_q_tiling_qk_swizzle_qkv" if vt else "flash_attn_mma_stages_split_q_tiling_qkv", *args, o, 2)
                    torch.cuda.synchronize()
                    _check(oracle, *args, o, vt=vt)
            finally:
                capi.tune("attn_d512", 0)
    else:
        with pytest.raises(capi.LcError) as e:   # the reference dispatcher stops at 256 for this entry
            capi.attn_call("flash_attn_mma_stages_split_q_tiling_qk_swizzle_qkv", q, k,
                           v.transpose(-2, -1).contiguous(), o, 2)
        assert e.value.status == capi.LC_ERR_HEADDIM


@pytest.mark.parametrize("D,N", [(512, 256), (256, 192), (512, 64)])
def test_bf16_large_head_dim(oracle, D, N):
    """BASELINE config 5: FFPA-style tiling at D = 512 in bfloat16 (extension; oracle on the bf16 inputs).
    bf16 has 8 mantissa bits: P and O round 8x coarser than fp16, hence the wider absolute band."""
    capi = _capi()
    B, H = 1, 2
    torch.manual_seed(D * 3 + N)
    q = torch.randn(B, H, N, D, device="cuda").to(torch.bfloat16)
    k = torch.randn(B, H, N, D, device="cuda").to(torch.bfloat16)
    v = torch.randn(B, H, N, D, device="cuda").to(torch.bfloat16)
    o = torch.full_like(q, float("nan"))
    capi.attn_fwd_bf16(q, k, v, o)
    torch.cuda.synchronize()
    truth = oracle.attn_bf16(q, k, v, B, H, N, D)
    d = np.abs(o.float().cpu().numpy() - truth)
    assert np.isfinite(d).all() and d.max() < 1.6e-2, d.max()
    assert d.mean() < 1.5e-3, d.mean()


def test_golden_fixtures(oracle, golden):
    capi = _capi()
    g = golden["attn"]
    for i in range(4):
        q, k, v = (torch.from_numpy(g[f"{n}{i}"].view(np.float16)).cuda() for n in ("q", "k", "v"))
        o = torch.zeros_like(q)
        capi.attn_fwd(q, k, v, o)
        torch.cuda.synchronize()
        d = np.abs(o.float().cpu().numpy() - g[f"o64_{i}"])     # unfused_standard_attn in fp64
        assert d.max() < tol.ATTN_MAX_ABS, d.max()


@pytest.mark.parametrize("nw", [0, 513, 515, 517, 514, 8])
def test_forced_rescale_spike(oracle, nw):
    """One K row matches one Q row so strongly that the running max jumps by >> 8 in the middle of the
    sequence (tile 5 of 8): every row's accumulator must be rescaled exactly once (rule 26)."""
    capi = _capi()
    B, H, N, D = 1, 2, 512, 128
    torch.manual_seed(42)
    q = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    k = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    v = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    k[:, :, 5 * 64 + 17] = 3.0 * q[:, :, 33]          # q33·k337 ~ 3*128 -> score ~ 34 after scaling
    k[:, :, 2 * 64 + 3] = 1.5 * q[:, :, 400]
    v[:, :, 5 * 64 + 17] = 7.0
    o = torch.zeros_like(q)
    capi.tune("attn_nw", nw)
    try:
        capi.attn_fwd(q, k, v, o)
        torch.cuda.synchronize()
    finally:
        capi.tune("attn_nw", 0)
    _check(oracle, q, k, v, o, max_abs=6e-3)
    assert abs(o[0, 0, 33].float().mean().item() - 7.0) < 0.02


def test_degenerate_inputs(oracle):
    capi = _capi()
    B, H, N, D = 1, 1, 256, 64
    ones = torch.ones(B, H, N, D, dtype=torch.half, device="cuda")       # --no-rand-qkv
    o = torch.zeros_like(ones)
    capi.attn_fwd(ones, ones, ones, o)
    torch.cuda.synchronize()
    assert (o.float() - 1).abs().max().item() < 1e-3
    kr = torch.ones_like(ones)                                            # --range-k: K rows = (i+1)/N
    for i in range(N):
        kr[:, :, i, :] = (i + 1) / N
    torch.manual_seed(1)
    v = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    capi.attn_fwd(ones, kr, v, o)
    torch.cuda.synchronize()
    _check(oracle, ones, kr, v, o)
    z = torch.zeros_like(ones)                                            # uniform softmax -> mean of V
    capi.attn_fwd(z, z, v, o)
    torch.cuda.synchronize()
    assert (o.float() - v.float().mean(dim=2, keepdim=True)).abs().max().item() < 1e-3


def test_every_reference_entry_name(oracle):
    capi = _capi()
    B, H, N, D = 1, 2, 256, 64
    torch.manual_seed(3)
    q = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    k = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    v = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    tv = v.transpose(-2, -1).contiguous()
    truth = oracle.attn(q, k, v, B, H, N, D, mode="f32")
    for name, fam, vt, acc, d2, d1, nargs in capi.attn_entries():
        for stages in (1, 2):
            o = torch.zeros_like(q)
            capi.attn_call(name, q, k, tv if vt else v, o, stages)
            torch.cuda.synchronize()
            d = np.abs(o.float().cpu().numpy() - truth).max()
            assert d < tol.ATTN_MAX_ABS, (name, stages, d)
    q512 = torch.zeros(1, 1, 64, 512, dtype=torch.half, device="cuda")
    with pytest.raises(capi.LcError) as e:
        capi.attn_call("flash_attn_mma_stages_split_q", q512, q512, q512, q512, 2)
    assert e.value.status == capi.LC_ERR_HEADDIM


def test_torch_extension_module_drop_in(oracle):
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "leetcuda_amd"))
    import flash_attn_lib
    B, H, N, D = 1, 4, 512, 128
    torch.manual_seed(8)
    q = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    k = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    v = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    for fn in (flash_attn_lib.flash_attn_mma_stages_split_q, flash_attn_lib.flash_attn_mma_stages_split_q_shared_qkv):
        o = torch.zeros_like(q)
        fn(q, k, v, o, 2)
        torch.cuda.synchronize()
        _check(oracle, q, k, v, o)
    o = torch.zeros_like(q)
    flash_attn_lib.flash_attn_cute(q, k, v, o)
    torch.cuda.synchronize()
    _check(oracle, q, k, v, o)
    with pytest.raises(RuntimeError, match="headdim not support!"):
        x = torch.zeros(1, 1, 64, 48, dtype=torch.half, device="cuda")
        flash_attn_lib.flash_attn_mma_stages_split_q(x, x, x, x, 2)


def test_full_size_config3_properties(oracle):
    """BASELINE config 3: B=4,H=32,S=4096,D=128 randn fp16 (flash_attn_mma.py:417-435) through the split-Q entry name, both `stages`:
    74 rows per sampled head (block seams + 64 random rows) x ALL keys under the N-scaled bound of tests/tol.py — 5e-4 + one output ulp at
    N = 4096, the bound configs 4 / 5a are held to (round-5 verdict: this test used a fixed 2e-3 on 7 rows) — and a spiked head."""
    capi = _capi()
    from tests.test_gpu_configs import _rows_for, _sampled_rows_check
    B, H, N, D = 4, 32, 4096, 128
    torch.manual_seed(0)
    q = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    k = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    v = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    # a spike late in the sequence on one head: forces the rescale path (scores of ~3 sqrt(D) against the planted key)
    k[3, 31, 3500] = 3.0 * q[3, 31, 33]
    v[3, 31, 3500] = 5.0
    rows = _rows_for(N, 3, extra=[33, 127, 128, 2047, 2048])
    o = torch.full_like(q, float("nan"))
    capi.attn_call("flash_attn_mma_stages_split_q", q, k, v, o, 2)
    torch.cuda.synchronize()
    assert torch.isfinite(o).all()
    _sampled_rows_check(oracle, q, k, v, o, [(0, 0), (1, 17), (2, 5), (3, 30)], rows, tol.ATTN_MAX_ABS)
    _sampled_rows_check(oracle, q, k, v, o, [(3, 31)], rows, rtol=tol.ATTN_RTOL_SPIKE)      # the spiked head
    o1 = torch.full_like(q, float("nan"))
    capi.attn_call("flash_attn_mma_stages_split_q", q, k, v, o1, 1)       # stages = 1: the same path (split_q.cu:778)
    torch.cuda.synchronize()
    assert torch.equal(o, o1)
    capi.attn_fwd(q, k, v, o1)                                             # ... and the C-ABI entry without a name
    torch.cuda.synchronize()
    assert torch.equal(o, o1)
    # V = const  =>  O = const for every one of the 524288 rows (softmax weights sum to one)
    vc = torch.full_like(v, 0.75)
    capi.attn_fwd(q, k, vc, o)
    torch.cuda.synchronize()
    assert (o.float() - 0.75).abs().max().item() < 1e-3
    # linearity in V: attn(q,k,v1+v2) == attn(q,k,v1) + attn(q,k,v2) on exactly representable values
    v1 = (torch.randint(-8, 9, v.shape, device="cuda").half() / 8)
    v2 = (torch.randint(-8, 9, v.shape, device="cuda").half() / 8)
    o1, o2, o12 = torch.zeros_like(q), torch.zeros_like(q), torch.zeros_like(q)
    capi.attn_fwd(q, k, v1, o1)
    capi.attn_fwd(q, k, v2, o2)
    capi.attn_fwd(q, k, v1 + v2, o12)
    torch.cuda.synchronize()
    assert (o12.float() - o1.float() - o2.float()).abs().max().item() < 2e-3


@pytest.mark.parametrize("nw", [513, 515, 517, 514, 8])
def test_scale_jumps_and_extreme_scores(oracle, nw):
    """The merged-phase kernel treats the running max as a mere SCALE and only corrects it when a half-tile's row sums
    get large (attn_w4u.hip).  Inputs that force that path in many places: (a) scores that grow steadily along the
    sequence (every tile raises the max), (b) a huge uniform score level (s ~ 1000 in log2 units), (c) a first tile far
    ABOVE everything else (later P underflow harmlessly), (d) maxima that jump by > 2^14 in the LAST half-tile."""
    capi = _capi()
    B, H, N, D = 1, 2, 1024, 128
    torch.manual_seed(11)
    q = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    v = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    cases = {}
    k = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    ramp = torch.linspace(0.0, 6.0, N, device="cuda").half()
    cases["ramp"] = (q, (k + ramp[None, None, :, None] * q[:, :, :1].sign()).contiguous())      # growing scores
    cases["level"] = (torch.full_like(q, 8.0), torch.full_like(q, 8.0))                       # s = 64*128/sqrt(128) ~ 724
    k2 = k.clone()
    k2[:, :, :32] = 4.0 * q[:, :, :32]                                                           # dominant first half-tile
    cases["first"] = (q, k2)
    k3 = k.clone()
    k3[:, :, N - 7] = 3.0 * q[:, :, 100]
    k3[:, :, N - 40] = 2.0 * q[:, :, 900]
    cases["last"] = (q, k3)
    capi.tune("attn_nw", nw)
    try:
        for name, (qq, kk) in cases.items():
            o = torch.full_like(q, float("nan"))
            capi.attn_fwd(qq, kk, v, o)
            torch.cuda.synchronize()
            truth = oracle.attn(qq, kk, v, B, H, N, D, mode="f32")
            d = np.abs(o.float().cpu().numpy() - truth)
            assert np.isfinite(d).all() and d.max() < 8e-3, (name, d.max())
    finally:
        capi.tune("attn_nw", 0)


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
@pytest.mark.parametrize("D", [256, 512])
def test_full_width_large_head_dim_kernel(oracle, D, dtype):
    """The full-width kernels (one workgroup owns all D columns: attn_bigd7 for D = 256, attn_bigd6 for D = 512; attn_bigd2 on the other
    MFMA shape under "attn_d512" = 3) against the oracle, against round 1's independently written column-split kernel (knob 1), and on
    inputs that force their rescale paths (the running max is only a scale there): a spike row late in the sequence and a dominant
    first tile."""
    capi = _capi()
    B, H, N = 1, 3, 1024
    bf = dtype == "bf16"
    tdt = torch.bfloat16 if bf else torch.half
    torch.manual_seed(D + (7 if bf else 0))
    q = torch.randn(B, H, N, D, device="cuda").to(tdt)
    k = torch.randn(B, H, N, D, device="cuda").to(tdt)
    v = torch.randn(B, H, N, D, device="cuda").to(tdt)
    k2 = k.clone()
    k2[:, :, 700] = 1.5 * q[:, :, 33]            # score ~ 1.5*D/sqrt(D): far above everything before it
    k2[:, :, :64] = 0.5 * q[:, :, :64]           # dominant first tile for the first 64 query rows
    tol_max = 1.6e-2 if bf else 4e-3
    run = (lambda a, b, c, o: capi.attn_fwd_bf16(a, b, c, o)) if bf else \
        (lambda a, b, c, o: capi.attn_call("flash_attn_mma_stages_split_q_tiling_qkv", a, b, c, o, 2))
    for kk in (k, k2):
        outs = []
        for knob in (0, 1, 2, 3, 4):   # 3: the other MFMA shape (attn_bigd2 where auto is attn_bigd7 / attn_bigd6); 4: attn_bigd7 on this small grid too
            capi.tune("attn_d512", knob)
            try:
                o = torch.full_like(q, float("nan"))
                run(q, kk, v, o)
                torch.cuda.synchronize()
            finally:
                capi.tune("attn_d512", 0)
            outs.append(o.float().cpu().numpy())
        truth = oracle.attn_bf16(q, kk, v, B, H, N, D) if bf else oracle.attn(q, kk, v, B, H, N, D, mode="f32")
        for o in outs:
            d = np.abs(o - truth)
            assert np.isfinite(d).all() and d.max() < tol_max, d.max()
        assert all(np.abs(outs[0] - o).max() < tol_max for o in outs[1:])
    assert capi.attn_kernel_name(N, D, False, bf).startswith("attn_fwd_bigd7_kernel" if D == 256 else "attn_fwd_bigd6_kernel")   # (on a grid that fills the GPU)


@pytest.mark.parametrize("D,N", [(128, 256), (256, 128)])
def test_every_entry_name_at_its_head_dim_limits(oracle, D, N):
    """Every reference entry name x stages {1, 2} at D = 128 and D = 256 — the boundary values of the wrappers' switch(d)
    (flash_attn_mma_share_kv.cu:869-921: shared_kv / shared_qkv take d <= 128 with stages 2 and d <= 256 with stages 1;
    split_q / split_kv stop at 128; tiling_* go to 1024; *_tiling_qk_swizzle_qkv stops at 256; flash_attn_cute at 256) —
    with V handed over as [B,H,D,N] for the three *_swizzle_qkv entries that take it that way."""
    capi = _capi()
    B, H = 1, 2
    torch.manual_seed(1000 + D)
    q = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    k = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    v = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    tv = v.transpose(-2, -1).contiguous()
    truth = oracle.attn(q, k, v, B, H, N, D, mode="f32")
    n_ok = n_rej = n_vt = 0
    for name, fam, vt, acc, d2, d1, nargs in capi.attn_entries():
        for stages in (1, 2):
            limit = d2 if (stages > 1 or nargs == 4) else d1
            o = torch.full_like(q, float("nan"))
            if D > limit:
                with pytest.raises(capi.LcError) as e:
                    capi.attn_call(name, q, k, tv if vt else v, o, stages)
                assert e.value.status == capi.LC_ERR_HEADDIM, (name, stages)
                assert torch.isnan(o).all()                       # nothing was launched
                n_rej += 1
                continue
            capi.attn_call(name, q, k, tv if vt else v, o, stages)
            torch.cuda.synchronize()
            d = np.abs(o.float().cpu().numpy() - truth).max()
            assert d < tol.ATTN_MAX_ABS, (name, stages, d)
            n_ok += 1
            n_vt += int(bool(vt))
    assert n_ok + n_rej == 2 * len(capi.attn_entries()) and n_vt >= 2
    if D == 256:
        assert n_rej >= 2 * 2 + 8      # split_kv / split_q at both stage counts; the shared_* entries with stages = 2


def test_non_finite_scores_take_the_slow_path(oracle):
    """the overflow guards must route NaN / inf row sums into the slow path by construction (bit-pattern
    compare, lc_common.h psum_below), not by what -fno-honor-nans lets the compiler do with !(a < b).  One K row of one
    head is +inf: that head's slow-path counter must tick with the non-finite flag, every OTHER head must stay exact."""
    capi = _capi()
    B, H, N, D = 1, 3, 512, 128
    torch.manual_seed(9)
    q = torch.randn(B, H, N, D, dtype=torch.half, device="cuda").abs()      # q > 0 so that q . (+inf row) = +inf, not NaN
    k = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    v = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    k[0, 1, 300] = float("inf")
    assert capi.attn_kernel_name(N, D).startswith("attn_fwd_w4u_kernel<128,false,1>")
    capi.attn_slowpath_stats(reset=True)
    o = torch.zeros_like(q)
    capi.tune("attn_split", 1)          # the merged-phase kernel itself (auto hands this 6-workgroup grid to the split-KV path: tested below)
    try:
        capi.attn_fwd(q, k, v, o)
        torch.cuda.synchronize()
    finally:
        capi.tune("attn_split", 0)
    st = capi.attn_slowpath_stats(reset=True)
    assert st[0] >= 1 and st[2] >= 1, st                     # executions, of which with a non-finite row sum
    good = [0, 2]
    truth = oracle.attn(q[:, good].contiguous(), k[:, good].contiguous(), v[:, good].contiguous(), B, 2, N, D, mode="f32")
    d = np.abs(o[:, good].float().cpu().numpy() - truth)
    assert np.isfinite(d).all() and d.max() < tol.ATTN_MAX_ABS
    # the poisoned head follows IEEE like the reference would (exp(inf - inf) = NaN in its rows): not finite, not silently wrong
    assert not torch.isfinite(o[0, 1]).all()


@pytest.mark.parametrize("D", [128, 64, 96, 32])
def test_generalised_and_generated_kernels_reproduce_the_reference_kernel_bit_for_bit(oracle, D):
    """attn_fwd_w4u_kernel (attn_w4u.hip: compiler-scheduled fillers between one-MFMA asm statements, 513) and attn_fwd_w4i_kernel
    (attn_w4i.hip: each phase ONE generated asm statement on reserved registers, uniform padded loop, 514) are two independently
    scheduled instruction streams of one arithmetic — same MFMA order per accumulator, same exp2 / row-sum / pack sequence — and must
    agree BIT FOR BIT at D = 128 and D = 64, on random data and on inputs that take the overflow slow path (spike rows, a dominant
    first half-tile, a growing ramp), with the slow-path counter confirming the path was taken."""
    capi = _capi()
    B, H, N = 2, 3, 2048
    torch.manual_seed(513 + D)
    q = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    k = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    v = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    k2 = k.clone()
    k2[:, :, 1500] = 4.0 * q[:, :, 33]
    k2[:, :, N - 3] = 4.0 * q[:, :, 700]
    k2[:, :, :32] = 3.0 * q[:, :, :32]
    ramp = torch.linspace(0.0, 8.0, N, device="cuda").half()
    k3 = (k + ramp[None, None, :, None] * q[:, :, :1].sign()).contiguous()
    kernels = (513, 514) if D in (128, 64) else (514,)   # D = 96 / 32: the two schedules of the generated kernel
    for ci, kk in enumerate((k, k2, k3)):
        outs = {}
        for nw, sched in [(k_, 0) for k_ in kernels] + [(514, 1)]:        # (514, 1): the generated kernel's second schedule
            capi.tune("attn_nw", nw)
            capi.tune("attn_w4i_sched", sched)
            try:
                want = {513: f"attn_fwd_w4u_kernel<{D},false,0>", 514: f"attn_fwd_w4i_kernel<{D},{sched}>"}[nw]
                assert capi.attn_kernel_name(N, D).startswith(want)
                capi.attn_slowpath_stats(reset=True)
                o = torch.full_like(q, float("nan"))
                capi.attn_fwd(q, kk, v, o)
                torch.cuda.synchronize()
                st = capi.attn_slowpath_stats(reset=True)
            finally:
                capi.tune("attn_nw", 0)
                capi.tune("attn_w4i_sched", 1)      # (the default)
            assert (st[0] > 0) == (ci > 0), (nw, sched, ci, st)
            outs[(nw, sched)] = o
        ref = outs[(kernels[0], 0)]
        for key, o in outs.items():
            assert torch.equal(ref, o), (D, ci, key)
        _check(oracle, q, kk, v, outs[(514, 1)], max_abs=8e-3)


@pytest.mark.parametrize("vt", [False, True], ids=["v_nd", "v_dn"])
@pytest.mark.parametrize("D", [128, 64])
@pytest.mark.parametrize("shape", [(1, 37, 2048), (3, 43, 1024), (2, 3, 2048), (1, 130, 512), (2, 48, 4096)])
def test_block_walks_compute_the_same_bits(oracle, D, shape, vt):
    """attn_fwd_w4u_kernel<D, VT, WALK> (attn_w4u.hip): WALK 0 = one 256-row query block per workgroup (513), 1 = one persistent
    workgroup per CU walking w, w + G, ... (515: the K / V tiles 0 / 1 and the Q rows of the NEXT block are fetched while the current
    one finishes, O is staged behind ring slots 0 / 1), 2 = the same with a dynamic per-XCD block queue (517: one atomic per block,
    claimed one block ahead, passed through an LDS mailbox; the claim counters reset themselves).  Inside a block the three are the
    same instructions, so the outputs must be IDENTICAL — with more blocks than CUs (296 ... 1536: a workgroup runs 1-6 blocks), fewer
    (48: every walk is the one-block launch), heads that take the overflow slow path right before / after a seam, from launch to launch
    (the dynamic walk's block -> workgroup assignment differs every time), and for both V layouts."""
    capi = _capi()
    B, H, N = shape
    torch.manual_seed(515 + D + N)
    q = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    k = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    v = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    vin = v.transpose(-2, -1).contiguous() if vt else v
    k2 = k.clone()
    k2[:, ::3, N - 3] = 4.0 * q[:, ::3, 300]          # last tile of every third head: slow path in the phases next to the seam
    k2[:, 1::5, :32] = 3.0 * q[:, 1::5, :32]          # dominant first half-tile: slow path right behind the seam
    vts = "true" if vt else "false"
    for ci, kk in enumerate((k, k2)):
        outs = {}
        for nw in (513, 515, 517, 517, 515, 517):
            capi.tune("attn_nw", nw)
            try:
                assert capi.attn_kernel_name(N, D, vt) == f"attn_fwd_w4u_kernel<{D},{vts},{(nw - 513) // 2}>"
                capi.attn_slowpath_stats(reset=True)
                o = torch.full_like(q, float("nan"))
                capi.attn_fwd(q, kk, vin, o, v_transposed=vt)
                torch.cuda.synchronize()
                st = capi.attn_slowpath_stats(reset=True)
            finally:
                capi.tune("attn_nw", 0)
            assert (st[0] > 0) == (ci > 0), (nw, ci, st)
            outs.setdefault(nw, []).append((o, st[0]))
        ref, ref_slow = outs[513][0]
        assert torch.isfinite(ref).all()
        for nw in (515, 517):
            for o, slow in outs[nw]:
                assert torch.equal(ref, o), (D, shape, ci, nw)
                assert slow == ref_slow, (D, shape, ci, nw, slow, ref_slow)      # the same half-tiles took the slow path
        if ci == 0 and H <= 43:
            _check(oracle, q[:, :2].contiguous(), kk[:, :2].contiguous(), vin[:, :2].contiguous(), ref[:, :2].contiguous(), vt=vt,
                   max_abs=8e-3)


@pytest.mark.parametrize("D", [128, 64])
def test_dynamic_walk_many_launches_leave_the_queue_clean(oracle, D):
    """The dynamic walk's claim counters live in 1024 rotating slots that every launch must leave zeroed (the last workgroup out
    resets its slot): 2100 back-to-back launches revisit every slot at least twice; every output must equal the one-block launch's and
    no row may stay unwritten (a dirty counter would make workgroups skip blocks: NaN rows in a NaN-prefilled output)."""
    capi = _capi()
    B, H, N = 1, 40, 2048                        # 320 blocks on 256 CUs: most workgroups claim one more block, all claim at least once
    torch.manual_seed(517 + D)
    q = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    k = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    v = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    ref = torch.full_like(q, float("nan"))
    capi.tune("attn_nw", 513)
    try:
        capi.attn_fwd(q, k, v, ref)
    finally:
        capi.tune("attn_nw", 0)
    torch.cuda.synchronize()
    assert torch.isfinite(ref).all()
    o = torch.full_like(q, float("nan"))
    capi.tune("attn_nw", 517)
    try:
        for it in range(2100):
            if it % 300 == 0 or it >= 2096:
                o.fill_(float("nan"))
                capi.attn_fwd(q, k, v, o)
                torch.cuda.synchronize()
                assert torch.equal(ref, o), it
            else:
                capi.attn_fwd(q, k, v, o)
    finally:
        capi.tune("attn_nw", 0)
    torch.cuda.synchronize()
    assert torch.equal(ref, o)
    # two streams at once: launches in flight together use different slots
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    o1, o2 = torch.full_like(q, float("nan")), torch.full_like(q, float("nan"))
    capi.tune("attn_nw", 517)
    try:
        for _ in range(20):
            with torch.cuda.stream(s1):
                capi.attn_fwd(q, k, v, o1)
            with torch.cuda.stream(s2):
                capi.attn_fwd(q, k, v, o2)
    finally:
        capi.tune("attn_nw", 0)
    torch.cuda.synchronize()
    assert torch.equal(ref, o1) and torch.equal(ref, o2)


@pytest.mark.parametrize("D", [64, 96, 32])
@pytest.mark.parametrize("nw", [0, 514, 8])
def test_scale_jumps_and_spikes_d64(oracle, nw, D):
    """The D = 64 instantiation of the merged-phase kernel (running max = a mere scale, corrected by the overflow slow path):
    the inputs of test_scale_jumps_and_extreme_scores / test_forced_rescale_spike at D = 64 and D = 96 (generated kernel only, on
    256-B padded LDS rows), against the lock-step kernel too."""
    capi = _capi()
    B, H, N = 1, 2, 1024
    torch.manual_seed(640 + D)
    q = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    k = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    v = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    cases = {"plain": (q, k)}
    ramp = torch.linspace(0.0, 8.0, N, device="cuda").half()
    cases["ramp"] = (q, (k + ramp[None, None, :, None] * q[:, :, :1].sign()).contiguous())
    cases["level"] = (torch.full_like(q, 8.0), torch.full_like(q, 8.0))              # s = 64 D / sqrt(D) = 512 at D = 64 (x log2 e)
    k2 = k.clone()
    k2[:, :, :32] = 5.0 * q[:, :, :32]
    cases["first"] = (q, k2)
    k3 = k.clone()
    k3[:, :, N - 7] = 4.0 * q[:, :, 100]
    k3[:, :, 5 * 64 + 17] = 4.0 * q[:, :, 33]
    k3[:, :, N - 40] = 3.0 * q[:, :, 900]
    cases["spikes"] = (q, k3)
    capi.tune("attn_nw", nw)
    try:
        capi.attn_slowpath_stats(reset=True)
        for name, (qq, kk) in cases.items():
            o = torch.full_like(q, float("nan"))
            capi.attn_fwd(qq, kk, v, o)
            torch.cuda.synchronize()
            truth = oracle.attn(qq, kk, v, B, H, N, D, mode="f32")
            d = np.abs(o.float().cpu().numpy() - truth)
            assert np.isfinite(d).all() and d.max() < 8e-3, (name, d.max())
        st = capi.attn_slowpath_stats(reset=True)
        assert (st[0] > 0) == (nw != 8), st          # the merged-phase kernels took their slow path on these inputs
    finally:
        capi.tune("attn_nw", 0)


# ---- split-KV (round 5): grids that do not fill the GPU ------------------------------------------------------------------------
# (B, H, N, D, S): S = the factor the auto rule picks on a 256-CU device (lc_abi.hip attn_split_auto's cost model), or — negative — a factor
# forced through lc_tune_set "attn_split" on a shape auto leaves alone (too few KV tiles for the combine to pay)
SPLIT_SHAPES = [(1, 8, 1024, 128, 4), (1, 8, 2048, 64, 4), (1, 16, 2048, 128, 2), (1, 5, 4096, 64, 2), (1, 3, 768, 128, 2), (2, 3, 512, 128, -2), (1, 1, 256, 64, -2)]


@pytest.mark.parametrize("vt", [False, True], ids=["v_nd", "v_dn"])
@pytest.mark.parametrize("shape", SPLIT_SHAPES, ids=[str(s) for s in SPLIT_SHAPES])
def test_split_kv_on_grids_that_do_not_fill_the_gpu(oracle, shape, vt):
    """Round-4 verdict (missing #2): the tuned kernels own 256 query rows per workgroup, so (1,8,1024,128) ran on 32 of 256 CUs.  Auto
    now launches attn_fwd_w4u_kernel<D, VT, 3>: S workgroups per query block over disjoint KV ranges (partials: normalised fp16 O + the
    base-2 log-sum-exp per row in the cached per-stream workspace) + attn_split_combine_kernel.  Against the oracle on random data, with a
    spike key planted in the LAST KV range and one in a middle range (the combine must weight ranges by 2^(L_s - L): a range holding a
    spike dominates its row), against the unsplit kernel, for both V layouts, and bit-reproducible from launch to launch."""
    capi = _capi()
    B, H, N, D, S = shape
    # the expected auto factors are those of a 256-CU device with the built-in constants: the RULE is asked to reason that way on any device
    # (round 6: "rule_cus" / "attn_calib"; the grids are still sized with the real CU count — a split that fits 256 CUs is correct anywhere)
    capi.tune("rule_cus", 256)
    capi.tune("attn_calib", 1)
    try:
        _split_kv_body(oracle, capi, shape, vt)
    finally:
        capi.tune("rule_cus", 0)
        capi.tune("attn_calib", 0)


def _split_kv_body(oracle, capi, shape, vt):
    B, H, N, D, S = shape
    torch.manual_seed(519 + N + D + B * H)
    q = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    k = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    v = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    k[:, :, N - 3] = 3.0 * q[:, :, 5]            # spike in the last range: query row 5 attends almost only to key N - 3
    k[:, :, N // 2 + 1] = 3.0 * q[:, :, 200]       # ... and one in a middle range
    vin = v.transpose(-2, -1).contiguous() if vt else v
    vts = "true" if vt else "false"
    if S < 0:
        assert capi.attn_kernel_name(N, D, vt, bh=B * H).startswith("attn_fwd_kernel<")      # auto: not worth a split -> the 128-row lock-step kernel on this small grid
        capi.tune("attn_split", -S)
    try:
        name = capi.attn_kernel_name(N, D, vt, bh=B * H)
        assert name == f"attn_fwd_w4u_kernel<{D},{vts},3>", name
        assert capi.attn_kernel_name(N, D, vt).endswith(",1>")       # no batch / head count: a grid that fills the GPU
        outs = []
        for _ in range(2):
            o = torch.full_like(q, float("nan"))
            capi.attn_fwd(q, k, vin, o, v_transposed=vt)
            torch.cuda.synchronize()
            outs.append(o)
    finally:
        capi.tune("attn_split", 0)
    assert torch.equal(outs[0], outs[1])           # same shape, same device: the same bits
    truth = oracle.attn(q, k, v, B, H, N, D, mode="f32")
    ok, mx, ex = tol.attn_close(outs[0].float().cpu().numpy(), truth, N, rtol=tol.ATTN_RTOL_SPIKE)
    assert ok, (mx, ex)
    assert abs(float(outs[0][0, 0, 5].float().abs().max()) - float(v[0, 0, N - 3].float().abs().max())) < 2e-2     # the spike row IS v[N - 3]
    capi.tune("attn_split", 1)
    try:
        o1 = torch.full_like(q, float("nan"))
        capi.attn_fwd(q, k, vin, o1, v_transposed=vt)
        torch.cuda.synchronize()
        assert capi.attn_kernel_name(N, D, vt, bh=B * H).endswith(",1>")
    finally:
        capi.tune("attn_split", 0)
    # split vs unsplit: the partials are fp16-rounded once more (|O_s| <= max |v|: 2^-11 relative to the partial, weights sum to 1)
    d = (outs[0].float() - o1.float()).abs()
    assert float(d.max()) <= 2.0 ** -9 * max(1.0, float(v.float().abs().max())), float(d.max())


@pytest.mark.parametrize("D", [128, 64])
@pytest.mark.parametrize("S", [2, 4, 8, 16])
def test_split_kv_every_factor(oracle, S, D):
    """lc_tune_set "attn_split" = S forces S KV ranges per query block on any grid: N = 2048 = 32 KV tiles -> 16 / 8 / 4 / 2 tiles per range
    (2 = the shortest walk the kernel's prologue / last-tile structure admits); a non-finite key poisons exactly its own head."""
    capi = _capi()
    B, H, N = 1, 6, 2048
    torch.manual_seed(S * 7 + D)
    q = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    k = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    v = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    k[0, 4, 1500] = float("inf")
    q[0, 4] = q[0, 4].abs()
    capi.tune("attn_split", S)
    try:
        assert capi.attn_kernel_name(N, D, bh=B * H) == f"attn_fwd_w4u_kernel<{D},false,3>"
        o = torch.full_like(q, float("nan"))
        capi.attn_fwd(q, k, v, o)
        torch.cuda.synchronize()
        capi.attn_slowpath_stats(reset=True)
    finally:
        capi.tune("attn_split", 0)
    good = [0, 1, 2, 3, 5]
    truth = oracle.attn(q[:, good].contiguous(), k[:, good].contiguous(), v[:, good].contiguous(), B, len(good), N, D, mode="f32")
    ok, mx, ex = tol.attn_close(o[:, good].float().cpu().numpy(), truth, N)
    assert ok, (S, mx, ex)
    assert not torch.isfinite(o[0, 4]).all()          # IEEE, like the unsplit kernel: not finite, not silently wrong


def test_split_kv_inside_graph_capture_falls_back(oracle):
    """The split needs the cached per-stream workspace (hipMalloc on first use): while the stream is being captured into a graph the launcher runs
    the one-block walk instead — a captured drop-in launch stays ONE kernel node, and the replay computes the same attention."""
    capi = _capi()
    B, H, N, D = 1, 4, 1024, 128
    torch.manual_seed(77)
    q = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    k = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    v = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    o = torch.zeros_like(q)
    capi.attn_fwd(q, k, v, o)                     # outside capture: the split path
    torch.cuda.synchronize()
    o_split = o.clone()
    capi.tune("attn_split", 1)                    # warm-up of the kernel the capture will fall back to (its LDS attribute is set on first use)
    try:
        capi.attn_fwd(q, k, v, o)
        torch.cuda.synchronize()
    finally:
        capi.tune("attn_split", 0)
    o.zero_()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        capi.attn_fwd(q, k, v, o)
    g.replay()
    torch.cuda.synchronize()
    truth = oracle.attn(q, k, v, B, H, N, D, mode="f32")
    ok, mx, _ = tol.attn_close(o.float().cpu().numpy(), truth, N)
    assert ok, mx
    assert float((o.float() - o_split.float()).abs().max()) <= 2.0 ** -9 * float(v.float().abs().max())


def test_split_kv_many_launches_on_two_streams(oracle):
    """The partials live in a cached workspace per (device, stream): 100 back-to-back launches of two alternating shapes on one stream
    (the second shape re-uses — and regrows — the first one's buffer) interleaved with launches of a third problem on a second stream;
    every output identical to the first of its kind."""
    capi = _capi()
    torch.manual_seed(2024)
    shapes = [(1, 8, 1024, 128), (1, 4, 2048, 64), (1, 6, 1024, 128)]
    data = []
    for B, H, N, D in shapes:
        q = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
        k = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
        v = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
        data.append((q, k, v, torch.zeros_like(q)))
    side = torch.cuda.Stream()
    capi.tune("attn_split", 4)
    try:
        first = []
        for q, k, v, o in data:
            capi.attn_fwd(q, k, v, o)
            torch.cuda.synchronize()
            first.append(o.clone())
            truth = oracle.attn(q, k, v, *q.shape, mode="f32")
            assert tol.attn_close(o.float().cpu().numpy(), truth, q.shape[2])[0]
        for it in range(100):
            for i, (q, k, v, o) in enumerate(data[:2]):
                o.fill_(float("nan"))
                capi.attn_fwd(q, k, v, o)
            with torch.cuda.stream(side):
                q, k, v, o = data[2]
                o.fill_(float("nan"))
                capi.attn_fwd(q, k, v, o)
            if it % 25 == 24:
                torch.cuda.synchronize()
                for i, (_, _, _, o) in enumerate(data):
                    assert torch.equal(o, first[i]), (it, i)
    finally:
        capi.tune("attn_split", 0)


@pytest.mark.parametrize("vt", [False, True], ids=["v_nd", "v_dn"])
@pytest.mark.parametrize("D", [128, 64])
@pytest.mark.parametrize("N", [1152, 1216, 1344, 4224])
def test_n_multiple_of_128_runs_the_merged_phase_kernel(oracle, N, D, vt):
    """Round-4 verdict (missing #3): N % 256 == 128 is legal in the reference (flash_attn_mma_share_qkv.cu:839) and used to fall to the
    lock-step kernel.  The merged-phase kernel takes it — and every other N % 64 == 0 (1216 = 4 x 256 + 192, 1344 = 5 x 256 + 64) — with
    one block per workgroup: the head's last query block is partly real, the waves behind N walk the KV tiles on a clamped copy of the
    last row and store nothing — so the rows behind the tensor must stay untouched (O is allocated with a guard band) and the last real
    rows must be exact."""
    capi = _capi()
    B, H = 2, 3
    torch.manual_seed(N + D)
    q = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    k = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    v = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    k[:, :, N - 1] = 3.0 * q[:, :, N - 1]          # the last real query row attends to the last key: both sit in the half block / last tile
    vin = v.transpose(-2, -1).contiguous() if vt else v
    vts = "true" if vt else "false"
    assert capi.attn_kernel_name(N, D, vt, bh=B * H) == f"attn_fwd_w4u_kernel<{D},{vts},0>"
    buf = torch.full((B * H * N * D + 256 * D,), 7.0, dtype=torch.half, device="cuda")       # 256 guard rows behind O
    o = buf[:B * H * N * D].view(B, H, N, D)
    capi.attn_fwd(q, k, vin, o, v_transposed=vt)
    torch.cuda.synchronize()
    assert (buf[B * H * N * D:] == 7.0).all()
    truth = oracle.attn(q, k, v, B, H, N, D, mode="f32")
    ok, mx, ex = tol.attn_close(o.float().cpu().numpy(), truth, N, rtol=tol.ATTN_RTOL_SPIKE)
    assert ok, (mx, ex)
    capi.tune("attn_nw", 4 if N % 128 == 0 else 2)   # the lock-step kernel it replaces: agreement to the output's rounding
    try:
        o2 = torch.zeros_like(q)
        capi.attn_fwd(q, k, vin, o2, v_transposed=vt)
        torch.cuda.synchronize()
    finally:
        capi.tune("attn_nw", 0)
    assert float((o.float() - o2.float()).abs().max()) < 2e-3


@pytest.mark.parametrize("D,N", [(1024, 512), (512, 1024)])
def test_bigd_block_map_knob_computes_the_same_bits(oracle, D, N):
    """lc_tune_set "attn_bigd_map": a head's query blocks dealt round-robin over the XCDs (2; auto for D = 1024) or every XCD owning consecutive
    ones (1; auto for D = 512) — the fabric-traffic A/B of DESIGN.md section 4.4; the arithmetic of a block does not change.  With the
    round-robin map the D = 1024 kernel also staggers its KV walk by XCD ("attn_bigd_stagger"): another summation order, so the maps are
    compared with the stagger off, and the default against the oracle."""
    capi = _capi()
    B, H = 1, 5
    torch.manual_seed(D + N)
    q = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    k = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    v = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")

    def run(m, stag):
        capi.tune("attn_bigd_map", m)
        capi.tune("attn_bigd_stagger", stag)
        try:
            o = torch.full_like(q, float("nan"))
            capi.attn_call("flash_attn_mma_stages_split_q_tiling_qkv", q, k, v, o, 2)
            torch.cuda.synchronize()
            return o
        finally:
            capi.tune("attn_bigd_map", 0)
            capi.tune("attn_bigd_stagger", 0)
    o1, o2 = run(1, 1), run(2, 1)
    assert torch.equal(o1, o2)
    _check(oracle, q, k, v, o2)
    od = run(0, 0)                       # the defaults
    _check(oracle, q, k, v, od)
    assert torch.equal(od, run(0, 0))    # launch to launch
    assert float((od.float() - o1.float()).abs().max()) < 1e-3
    if D == 1024:
        assert torch.equal(od, run(2, 2)) and not torch.equal(od, o2)     # auto = round-robin + stagger: a different summation order



def test_split_kv_against_wave_quantisation(oracle):
    """g = 320 query blocks on 256 CUs are 1.25 rounds and cost 2; with 4 KV ranges per block the launch runs 5 rounds of a quarter of
    the walk (lc_abi.hip attn_split_auto: + 22 % at (1,10,8192,128), profiles/r5f_attn_split_quant.log).  The shape the rule picks it for,
    sampled rows x all keys against the oracle, and the unsplit kernel on the same inputs."""
    from tests.test_gpu_configs import _rows_for, _sampled_rows_check
    capi = _capi()
    capi.tune("rule_cus", 256)      # the rule's rounds are those of a 256-CU device: asked for explicitly (round 6), not skipped elsewhere
    capi.tune("attn_calib", 1)
    try:
        _split_quantisation_body(oracle, capi, _rows_for, _sampled_rows_check)
    finally:
        capi.tune("rule_cus", 0)
        capi.tune("attn_calib", 0)


def _split_quantisation_body(oracle, capi, _rows_for, _sampled_rows_check):
    B, H, N, D = 1, 10, 8192, 64
    torch.manual_seed(320)
    q = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    k = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    v = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    assert capi.attn_kernel_name(N, D, bh=B * H) == "attn_fwd_w4u_kernel<64,false,3>"
    o = torch.full_like(q, float("nan"))
    capi.attn_fwd(q, k, v, o)
    torch.cuda.synchronize()
    assert torch.isfinite(o).all()
    _sampled_rows_check(oracle, q, k, v, o, [(0, 0), (0, 4), (0, 9)], _rows_for(N, 256))
    capi.tune("attn_split", 1)
    try:
        o1 = torch.full_like(q, float("nan"))
        capi.attn_fwd(q, k, v, o1)
        torch.cuda.synchronize()
    finally:
        capi.tune("attn_split", 0)
    assert float((o.float() - o1.float()).abs().max()) <= 2.0 ** -9 * float(v.float().abs().max())


def test_random_shapes_through_every_dispatch_path(oracle):
    """Sixteen random (B, H, N, D, V layout) with N a multiple of 64 and D in {64, 128}: whatever the dispatcher picks — merged-phase kernel
    in any walk, split-KV, the ragged-N form, the lock-step kernel on small grids or odd N — must match the oracle; the kernel names seen
    are collected so that a change of the rules shows up here."""
    capi = _capi()
    rng = np.random.default_rng(5)
    seen = set()
    for _ in range(16):
        D = int(rng.choice([64, 128]))
        N = int(rng.integers(1, 41)) * 64
        bh = int(min(rng.integers(1, 41), max(1, 60_000_000 // (N * N))))
        B = 2 if bh % 2 == 0 and rng.random() < 0.5 else 1
        H = bh // B
        vt = bool(rng.random() < 0.3)
        torch.manual_seed(N + D + bh)
        q = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
        k = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
        v = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
        k[:, :, N - 1] = 2.5 * q[:, :, 0]
        vin = v.transpose(-2, -1).contiguous() if vt else v
        o = torch.full_like(q, float("nan"))
        capi.attn_fwd(q, k, vin, o, v_transposed=vt)
        torch.cuda.synchronize()
        name = capi.attn_kernel_name(N, D, vt, bh=B * H)
        seen.add(name.split("<")[0] + ("<..,3>" if name.endswith(",3>") else ""))
        truth = oracle.attn(q, k, v, B, H, N, D, mode="f32")
        ok, mx, ex = tol.attn_close(o.float().cpu().numpy(), truth, N, rtol=tol.ATTN_RTOL_SPIKE)
        assert ok, (B, H, N, D, vt, name, mx, ex)
    assert {"attn_fwd_kernel", "attn_fwd_w4u_kernel"} <= seen, seen


@pytest.mark.parametrize("D", [128, 64])
def test_dispatch_paths_agree_with_each_other(oracle, D):
    """The batch-variance contract of include/lc_abi.h made testable (round-5 verdict weak #2): which kernel serves a (b, h) problem depends
    on B x H and the CU count — lock-step kernel, merged-phase kernel, split-KV with S ranges + combine — and their low bits differ (summation
    order, the split's second fp16 rounding).  On ONE input the paths must agree with EACH OTHER within the roundings two correct kernels
    are entitled to (2^-10 |O| + 1.5 x 2^-10 E_p|v| + 1e-4, see below): tighter than the bound each is held to against the oracle
    (tests/tol.py: 1e-3 + 2^-10 |truth| at N = 1024), so a path that drifts inside the oracle bound still fails here.  A spiked key (late, inside the last KV range of every split) forces the rescale path in each."""
    capi = _capi()
    B, H, N = 1, 4, 1024
    torch.manual_seed(77 + D)
    q = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    k = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    v = torch.randn(B, H, N, D, dtype=torch.half, device="cuda")
    k[0, 1, 900] = 2.5 * q[0, 1, 17]
    v[0, 1, 900] = 4.0
    outs = {}

    def run(label, **knobs):
        for kn, val in knobs.items():
            capi.tune(kn, val)
        try:
            o = torch.full_like(q, float("nan"))
            capi.attn_fwd(q, k, v, o)
            torch.cuda.synchronize()
            outs[label] = (o.float().cpu().numpy(), capi.attn_kernel_name(N, D, False, bh=B * H))
        finally:
            for kn in knobs:
                capi.tune(kn, 0)

    run("lockstep8", attn_nw=8)
    run("lockstep4", attn_nw=4)
    run("merged", attn_nw=513, attn_split=1)
    run("split2", attn_split=2)
    run("split4", attn_split=4)
    run("auto")
    assert outs["lockstep8"][1].startswith("attn_fwd_kernel<"), outs["lockstep8"][1]
    assert outs["merged"][1].startswith("attn_fwd_w4u_kernel<") and outs["merged"][1].endswith(",0>"), outs["merged"][1]
    assert outs["split2"][1].endswith(",3>") and outs["split4"][1].endswith(",3>"), (outs["split2"][1], outs["split4"][1])
    truth = oracle.attn(q, k, v, B, H, N, D, mode="f32")
    for label, (o, name) in outs.items():
        ok, mx, ex = tol.attn_close(o, truth, N, rtol=tol.ATTN_RTOL_SPIKE)
        assert ok, (label, name, mx, ex)
    # what two correct kernels may differ by: both round P to fp16 (relative 2^-11 each, against different running maxima) — an error of
    # 2^-11 E_p|v| per path, E_p|v| = attention(q, k, |v|) — the split paths round their normalised partials once more (<= 2^-11 E_p|v|),
    # and everybody rounds O once (2^-11 |O| each)
    epv = oracle.attn(q, k, v.abs(), B, H, N, D, mode="f32")
    labels = sorted(outs)
    for i, a in enumerate(labels):
        for b in labels[i + 1:]:
            oa, ob = outs[a][0], outs[b][0]
            d = np.abs(oa - ob)
            bound = 2.0 ** -10 * np.maximum(np.abs(oa), np.abs(ob)) + 1.5 * 2.0 ** -10 * epv + 1e-4
            assert (d <= bound).all(), (a, b, outs[a][1], outs[b][1], float(d.max()), float((d - bound).max()))


@pytest.mark.parametrize("mode", ["fp16", "fp16_vt", "bf16"])
def test_d256_n_multiple_of_128_runs_the_ring_kernel(oracle, mode):
    """D = 256 with N % 256 == 128 (legal in the reference: its share_kv / tiling entries need N % 128 == 0): attn_bigd7 — 64 query rows
    per wave, 256 per workgroup — with the head's last block half real (waves 2 / 3 on a clamped copy of the last row, nothing stored)
    instead of attn_bigd2's 128-row workgroups (- 22 ... 28 %).  Guard band behind O, last real rows exact, both V layouts and bf16."""
    capi = _capi()
    B, H, N, D = 1, 3, 1408, 256
    vt, bf = mode == "fp16_vt", mode == "bf16"
    torch.manual_seed(1408 + len(mode))
    dt = torch.bfloat16 if bf else torch.half
    q = torch.randn(B, H, N, D, device="cuda").to(dt)
    k = torch.randn(B, H, N, D, device="cuda").to(dt)
    v = torch.randn(B, H, N, D, device="cuda").to(dt)
    k[:, :, N - 1] = (2.0 * q[:, :, N - 1].float()).to(dt)
    capi.tune("attn_d512", 4)        # (auto hands this 18-workgroup grid to attn_bigd2; 4 = attn_bigd7 on any grid)
    try:
        assert capi.attn_kernel_name(N, D, vt, bf, bh=B * H) == f"attn_fwd_bigd7_kernel<{'true' if bf else 'false'},{'true' if vt else 'false'}>"
        buf = torch.full((B * H * N * D + 256 * D,), 7.0, dtype=dt, device="cuda")
        o = buf[:B * H * N * D].view(B, H, N, D)
        if bf:
            capi.attn_fwd_bf16(q, k, v, o)
        else:
            capi.attn_fwd(q, k, v.transpose(-2, -1).contiguous() if vt else v, o, v_transposed=vt, family=capi.ATTN_TILING_QKV)
        torch.cuda.synchronize()
    finally:
        capi.tune("attn_d512", 0)
    assert (buf[B * H * N * D:].float() == 7.0).all()
    truth = oracle.attn_bf16(q, k, v, B, H, N, D) if bf else oracle.attn(q, k, v, B, H, N, D, mode="f32")
    ok, mx, ex = tol.attn_close(o.float().cpu().numpy(), truth, N, bf16=bf, rtol=(2.0 ** -6 if bf else tol.ATTN_RTOL_SPIKE))
    assert ok, (mode, mx, ex)


def test_split_rule_calibration_on_this_device():
    """Round-5 verdict (weak #13 / next #7): the split-KV cost model's constants are measured on the device (lc_tune_calibrate) instead of
    being the ones fitted on round 5's boxes.  The measurement must land near them on an MI355X (the call refuses anything outside
    [0.4, 2.5] x), the rule must answer with them, and "attn_calib" = 1 must restore the built-in answer."""
    capi = _capi()
    c = capi.tune_calibrate()
    assert c["adopted"], c
    assert 0.54 < c["tau128_us"] < 3.4 and 0.34 < c["tau64_us"] < 2.2 and 2.0 < c["x0_us"] < 12.5 and 1.0e6 < c["bytes_per_us"] < 6.5e6, c
    assert c["tau64_us"] < c["tau128_us"]
    names = {}
    for calib in (0, 1):
        capi.tune("attn_calib", calib)
        try:
            names[calib] = [capi.attn_kernel_name(N, D, bh=bh) for (bh, N, D) in ((8, 1024, 128), (4, 4096, 128), (2, 8192, 128), (128, 4096, 128), (1024, 8192, 128))]
        finally:
            capi.tune("attn_calib", 0)
    for n in names.values():
        assert n[0].endswith(",3>") and n[1].endswith(",3>") and n[2].endswith(",3>")          # grids far below a round: split with either set
        assert n[3].endswith(",1>") and n[4].endswith(",0>")                                    # config 3 / config 4: never


def test_graph_capture_first_use_of_the_fallback_kernels_in_a_fresh_process():
    """Round-5 advisor: the capture tests above warm the fallback kernel up BEFORE capturing; the realistic sequence is an eager warm-up that
    runs the workspace path (split-KV; split-K border strips) and a capture in which the workspace-free kernel is used for the FIRST time —
    its dynamic-LDS attribute (hipFuncSetAttribute, lc_launch.h set_dyn_lds) is then set while the stream is capturing.  A fresh
    interpreter, shapes no other test touches first: the captured call must be legal, replay must compute the attention / the GEMM."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    code = r'''
import sys
sys.path.insert(0, %r)
import torch
from leetcuda_amd import capi, host
capi.load()
torch.manual_seed(5)
B, H, N, D = 1, 6, 1024, 64
q = torch.randn(B, H, N, D, dtype=torch.half, device="cuda"); k = torch.randn_like(q); v = torch.randn_like(q)
vt = v.transpose(-2, -1).contiguous()
o = torch.zeros_like(q)
assert capi.attn_kernel_name(N, D, True, bh=B * H).endswith(",3>")
capi.attn_fwd(q, k, vt, o, v_transposed=True)          # eager: the split path (workspace allocated here)
torch.cuda.synchronize()
eager = o.clone(); o.zero_()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    capi.attn_fwd(q, k, vt, o, v_transposed=True)      # capture: attn_fwd_w4u_kernel<64,true,0> for the first time in this process
g.replay(); torch.cuda.synchronize()
ref = torch.nn.functional.scaled_dot_product_attention(q.float(), k.float(), v.float())
assert (o.float() - ref).abs().max().item() < 2e-3 and (o.float() - eager.float()).abs().max().item() < 2e-3
M, Nn, K = 640, 384, 2080
a = torch.randn(M, K, dtype=torch.half, device="cuda"); b = torch.randn(K, Nn, dtype=torch.half, device="cuda")
bb = host.as_col_major(b); c = torch.zeros(M, Nn, dtype=torch.half, device="cuda")
capi.hgemm(a, bb, c, layout=capi.LAYOUT_TN, variant=capi.HGEMM_MFMA256W4Y, swizzle_stride=256)   # eager: border strips with split-K
torch.cuda.synchronize()
ce = c.clone(); c.zero_()
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    capi.hgemm(a, bb, c, layout=capi.LAYOUT_TN, variant=capi.HGEMM_MFMA256W4Y, swizzle_stride=256)   # capture: the unsplit border form, first use
g2.replay(); torch.cuda.synchronize()
want = a.float() @ b.float()
assert (c.float() - want).abs().max().item() < 0.25 and (c.float() - ce.float()).abs().max().item() <= 0.125
print("captured ok")
''' % str(root)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "captured ok" in p.stdout, (p.stdout[-500:], p.stderr[-3000:])
