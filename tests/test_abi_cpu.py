"""CPU tests of the boundary: the C-ABI library loads, exports every symbol include/lc_abi.h declares,
rejects bad arguments without touching a GPU, and the two drop-in PyTorch modules export the reference
names and refuse CPU tensors loudly (there is no CPU path)."""
import ctypes as C
import re
import subprocess
import sys
from pathlib import Path

import pytest
import torch

from tests.test_host import ATTN_NAMES, HGEMM_NAMES

ROOT = Path(__file__).resolve().parent.parent


def _declared_symbols():
    txt = (ROOT / "include" / "lc_abi.h").read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lc_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_all_exported(built):
    from leetcuda_amd import capi
    lib = capi.load()
    declared = _declared_symbols()
    assert declared == sorted(capi.SYMBOLS)          # the ctypes table mirrors the header exactly
    for s in declared:
        assert getattr(lib, s) is not None
    nm = subprocess.run(["nm", "-D", "--defined-only", str(built["abi"])], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (lc_\w+)", nm))
    assert set(declared) <= exported
    assert lib.lc_abi_version() == 1


def test_status_strings_and_argument_errors(built):
    from leetcuda_amd import capi
    lib = capi.load()
    assert capi.status_string(capi.LC_ERR_HEADDIM) == "headdim not support!"     # split_q.cu:793
    assert capi.status_string(capi.LC_ERR_SHAPE) == "Tensor size mismatch!"      # hgemm_mma_stage.cu:2055
    # argument validation happens before any HIP call
    assert lib.lc_hgemm_f16(None, None, None, 256, 256, 256, 0, 0, 2, 1, None) == capi.LC_ERR_ARG
    one = C.c_void_p(16)
    assert lib.lc_hgemm_f16(one, one, one, 256, 256, 256, 7, 0, 2, 1, None) == capi.LC_ERR_ARG
    assert lib.lc_hgemm_f16(one, one, one, 0, 256, 256, 0, 0, 2, 1, None) == capi.LC_ERR_SHAPE
    assert lib.lc_hgemm_f16(one, one, one, 128, 256, 256, 0, capi.HGEMM_MFMA256, 2, 1, None) == capi.LC_ERR_SHAPE
    assert lib.lc_hgemm_call(b"no_such_entry", one, one, one, 256, 256, 256, 2, 0, 1, None) == capi.LC_ERR_ARG
    assert lib.lc_attn_fwd_f16(one, one, one, one, 1, 1, 100, 64, 0, 0, 0, 2, None) == capi.LC_ERR_SHAPE
    assert lib.lc_attn_fwd_f16(one, one, one, one, 1, 1, 128, 64, 0, 99, 0, 2, None) == capi.LC_ERR_ARG
    # head-dim limits of the reference dispatchers (split_q: 128; share_qkv stage2: 128, stage1: 256)
    assert lib.lc_attn_call(b"flash_attn_mma_stages_split_q", one, one, one, one, 1, 1, 128, 256, 2, None) \
        == capi.LC_ERR_HEADDIM
    assert lib.lc_attn_call(b"flash_attn_mma_stages_split_q_shared_qkv", one, one, one, one, 1, 1, 128, 256, 2,
                            None) == capi.LC_ERR_HEADDIM
    assert lib.lc_hgemm_vendor_f16(one, one, one, 256, 256, 256, 0, None) == capi.LC_ERR_VENDOR  # no init


def test_python_wrappers_refuse_cpu_tensors(built):
    from leetcuda_amd import capi
    a = torch.zeros(256, 256, dtype=torch.half)
    with pytest.raises(RuntimeError, match="no CPU path"):
        capi.hgemm(a, a, a)
    q = torch.zeros(1, 1, 64, 64, dtype=torch.half)
    with pytest.raises(RuntimeError, match="no CPU path"):
        capi.attn_fwd(q, q, q, q)


@pytest.fixture(scope="module")
def torch_mods():
    from leetcuda_amd import build
    build.build_torch_ext()
    sys.path.insert(0, str(ROOT / "leetcuda_amd"))
    import flash_attn_lib
    import toy_hgemm
    return toy_hgemm, flash_attn_lib


def test_torch_modules_export_reference_names(torch_mods):
    toy_hgemm, flash_attn_lib = torch_mods
    for n in HGEMM_NAMES:
        assert callable(getattr(toy_hgemm, n)), n
    for n in ATTN_NAMES:
        assert callable(getattr(flash_attn_lib, n)), n


def test_torch_modules_error_conventions(torch_mods, capfd):
    toy_hgemm, flash_attn_lib = torch_mods
    a = torch.zeros(256, 256, dtype=torch.half)
    with pytest.raises(RuntimeError, match="values must be torch::kHalf"):
        toy_hgemm.hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem(a.float(), a, a, 2, False, 1)
    assert "Tensor Info:" in capfd.readouterr().out            # the reference prints options first
    with pytest.raises(RuntimeError, match="no CPU path"):
        toy_hgemm.hgemm_naive_f16(a, a, a)
    with pytest.raises(TypeError):                             # 3-arg entry called with 6 args
        toy_hgemm.hgemm_naive_f16(a, a, a, 2, False, 1)
    q = torch.zeros(1, 1, 64, 64, dtype=torch.half)
    with pytest.raises(RuntimeError, match="values must be torch::kHalf"):
        flash_attn_lib.flash_attn_mma_stages_split_q(q.float(), q, q, q, 2)
    with pytest.raises(RuntimeError, match="no CPU path"):
        flash_attn_lib.flash_attn_cute(q, q, q, q)
