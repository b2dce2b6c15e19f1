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
