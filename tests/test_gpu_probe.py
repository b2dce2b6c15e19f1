"""GPU tests of the hardware lane maps the kernels are built on (MFMA operand / accumulator layouts and
the ds_read_b64_tr_b16 transpose). Transpose-detecting: A = I against an ASYMMETRIC B."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    """capi + liblc_diag.so (the probes of include/lc_diag.h live outside the drop-in library)."""
    from leetcuda_amd import capi
    capi.load()
    return capi, capi.load_diag()


def test_device_is_gfx950():
    capi, _ = _lib()
    assert capi.device_check() >= 200     # MI355X: 256 CUs


@pytest.mark.parametrize("shape", ["16", "32"])
def test_mfma_layouts(shape):
    capi, lib = _lib()
    R, Kd = (16, 32) if shape == "16" else (32, 16)
    g = torch.Generator().manual_seed(7)
    a = torch.randint(-4, 5, (R, Kd), generator=g).half()
    b = torch.randint(-4, 5, (R, Kd), generator=g).half()        # b[col][k]
    eye = torch.zeros(R, Kd).half()
    for i in range(min(R, Kd)):
        eye[i, i] = 1
    asym = torch.arange(R * Kd).reshape(R, Kd).half() / 8         # asymmetric, exact in fp16
    fn = lib.lc_probe_mfma16 if shape == "16" else lib.lc_probe_mfma32
    for x, y in ((a, b), (eye, asym), (asym, eye)):
        d = torch.zeros(R, R, dtype=torch.float32, device="cuda")
        xa, ya = x.cuda(), y.cuda()
        capi.check(fn(xa.data_ptr(), ya.data_ptr(), d.data_ptr(), None), "probe")
        torch.cuda.synchronize()
        want = x.float() @ y.float().t()                          # d[row][col] = sum_k a[row][k] b[col][k]
        assert torch.equal(d.cpu(), want)


def test_tr16_transpose_semantics():
    """Within each 16-lane group: lane i supplies row i>>2, cols 4*(i&3).. of a 4x16 block and receives
    column i (4 rows) — the map hgemm NN B-fragments and attention V-fragments rely on."""
    capi, lib = _lib()
    src = torch.arange(256, dtype=torch.int16)
    dst = torch.zeros(256, dtype=torch.int16, device="cuda")
    s = src.cuda()
    capi.check(lib.lc_probe_tr16(s.data_ptr(), dst.data_ptr(), None), "probe_tr16")
    torch.cuda.synchronize()
    got = dst.cpu().numpy().reshape(64, 4)
    mem = src.numpy().reshape(4, 16, 4)             # [group][lane][elem] as stored lane-linearly
    want = np.zeros((64, 4), np.int16)
    for g in range(4):
        block = mem[g].reshape(4, 16)               # lane i = row i>>2, cols 4*(i&3)..  ->  4x16 row-major
        for i in range(16):
            want[g * 16 + i] = block[:, i]
    assert (got == want).all(), (got[:20], want[:20])


def test_issue_probes_run_and_report_sane_cycle_counts():
    """tools/attn_mix_probe.py's entry points (include/lc_diag.h): one wave per SIMD or two next to the softmax share of an MFMA slot,
    and the MFMA operand register-file forms.  Sanity only — argument checks, and cycle counts between the matrix core's rate
    (16 cycles per v_mfma_f32_16x16x32_f16) and a generous upper bound; the numbers themselves are evidence (profiles/r3ab), not a test."""
    capi, lib = _lib()
    out = torch.zeros(16, dtype=torch.int64, device="cuda")
    assert lib.lc_probe_attn_mix(5, 0, out.data_ptr(), None) != 0
    assert lib.lc_probe_attn_mix(4, 9, out.data_ptr(), None) != 0
    assert lib.lc_probe_mfma_form(7, out.data_ptr(), None) != 0
    for waves in (4, 8):
        for mix in (0, 3, 4):
            out.zero_()
            capi.check(lib.lc_probe_attn_mix(waves, mix, out.data_ptr(), None), "probe")
            torch.cuda.synchronize()
            per_simd = out.cpu().numpy()[:waves].max() / 2048 / (waves // 4)
            assert 15.5 <= per_simd <= 80.0, (waves, mix, per_simd)
    for form in range(5):
        out.zero_()
        capi.check(lib.lc_probe_mfma_form(form, out.data_ptr(), None), "probe")
        torch.cuda.synchronize()
        per_mfma = out.cpu().numpy()[:4].max() / 4096
        assert 15.5 <= per_mfma <= 24.0, (form, per_mfma)
