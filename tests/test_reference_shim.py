"""CPU test of tools/run_reference_bench.py: the reference's own bench scripts run UNMODIFIED from
their own directory; the shim's three interceptions (cpp_extension.load, `import toy_hgemm`, the flash_attn stub) are
hit, and a GPU-less run proceeds exactly up to the script's first CUDA use.  Needs /root/reference (skipped on the
GPU box, where the driver does not mount it)."""
import hashlib
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
REF = Path("/root/reference/kernels")
pytestmark = pytest.mark.skipif(not REF.exists(), reason="reference not mounted")


def _run(script, *args):
    before = hashlib.sha256(script.read_bytes()).hexdigest()
    env = dict(os.environ, LC_SHIM_DRYRUN="1")
    p = subprocess.run([sys.executable, str(ROOT / "tools" / "run_reference_bench.py"), str(script), *args],
                       capture_output=True, text=True, timeout=300, env=env, cwd="/tmp")
    assert hashlib.sha256(script.read_bytes()).hexdigest() == before        # source untouched
    return p


def test_flash_attn_bench_reaches_first_cuda_allocation(built):
    from leetcuda_amd import build
    build.build_torch_ext()
    p = _run(REF / "flash-attn" / "flash_attn_mma.py", "--B", "1", "--H", "2", "--N", "256", "--D", "64")
    out = p.stdout + p.stderr
    assert "sys.modules['flash_attn'] = stub" in out
    assert "cpp_extension.load(name='flash_attn_lib'" in out and "-> prebuilt leetcuda_amd/flash_attn_lib" in out
    # ... and the script got past its build + argument handling to its first device="cuda" tensor
    assert "in get_qkvo" in out and "No HIP GPUs are available" in out and p.returncode != 0


def test_hgemm_bench_imports_toy_hgemm_and_force_build_is_intercepted(built):
    from leetcuda_amd import build
    build.build_torch_ext()
    script = REF / "hgemm" / "hgemm.py"
    p = _run(script, "--M", "256", "--N", "256", "--K", "256")
    out = p.stdout + p.stderr
    assert "Import toy-hgemm library done, use it!" in out            # kernels/hgemm/tools/utils.py:131
    assert "No HIP GPUs are available" in out
    p = _run(script, "--M", "256", "--N", "256", "--K", "256", "--force-build")
    out = p.stdout + p.stderr
    assert "Force hgemm lib build from sources" in out                 # tools/utils.py:143
    assert "cpp_extension.load(name='hgemm_lib'" in out and "-> prebuilt leetcuda_amd/toy_hgemm" in out
    assert "No HIP GPUs are available" in out
