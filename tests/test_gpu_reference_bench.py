"""f1 / f4 on the MI355X.

f1: the reference's UNMODIFIED bench scripts executed against this library through tools/run_reference_bench.py.  The
    scripts are not part of this repository: tools/stage_reference.sh copies them byte for byte into the git-ignored
    _refstage/ for one gpurun call; without that directory (the driver's round-end box) the f1 tests skip and the logs
    committed under profiles/ are the record.
f4: tools/cpp/hgemm_bench.bin, the torch-free C++ bench + error-check harness (reference: the main() at the tail of
    kernels/hgemm/mma/basic/hgemm_mma_stage.cu:1965-2038 + kernels/hgemm/utils/utils.h:7-91,217-277): every size up to
    --max-n must PASS the thresholded check against hipBLASLt, NN and TN."""
import os
import re
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
STAGE = ROOT / "_refstage" / "kernels"
needs_stage = pytest.mark.skipif(not (STAGE / "hgemm" / "hgemm.py").exists(),
                                 reason="reference scripts not staged (tools/stage_reference.sh; never committed)")


def _shim(script, *args, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MPLBACKEND="Agg")
    p = subprocess.run([sys.executable, str(ROOT / "tools" / "run_reference_bench.py"), str(script), *args],
                       capture_output=True, text=True, timeout=timeout, env=env, cwd=str(ROOT))
    return p.returncode, p.stdout + p.stderr


@needs_stage
def test_f1_hgemm_py_unmodified_runs_every_family():
    rc, out = _shim(STAGE / "hgemm" / "hgemm.py", "--mma-all", "--wmma-all", "--cuda-all", "--mma-tn", "--cute-tn",
                    "--MNK", "2048", "--show-all-info", "--sleep", "0.01")
    assert rc == 0, out[-3000:]
    assert "Import toy-hgemm library done, use it!" in out
    rows = re.findall(r"^\s*(\S+): \[.*TFLOPS: ([\d.]+)", out, flags=re.M)
    tags = {t for t, _ in rows}
    # the three kernel families + both vendor lines printed a TFLOPS row (hgemm.py:287-300)
    assert any("mma2x4+warp4x4x2" in t for t in tags) and any("wmma" in t for t in tags) and any("f16x8" in t for t in tags), tags
    assert "(cublas)" in tags and "tn(cublas)" in tags and any("cute" in t for t in tags), tags
    assert len(rows) >= 35 and all(float(v) > 0.5 for _, v in rows)


@needs_stage
def test_f1_flash_attn_mma_py_check_passes():
    rc, out = _shim(STAGE / "flash-attn" / "flash_attn_mma.py", "--B", "1", "--H", "8", "--N", "1024", "--D", "64",
                    "--check", "--show-all", "--others", "--seed", "1")
    assert rc == 0, out[-3000:]
    assert "cpp_extension.load(name='flash_attn_lib'" in out
    checks = re.findall(r"all close: (\w+)", out)          # flash_attn_mma.py:476-492 prints one line per kernel
    assert len(checks) >= 20 and all(c == "True" for c in checks), (len(checks), [c for c in checks if c != "True"][:3])


@pytest.mark.parametrize("layout", ["nn", "tn"])
def test_f4_cpp_bench_error_check_passes(layout):
    from leetcuda_amd import build
    exe = build.build_cpp_bench()
    p = subprocess.run([str(exe), "--layout", layout, "--max-n", "1280", "--check-n", "5", "--outer", "2"], capture_output=True,
                       text=True, timeout=600, cwd=str(ROOT))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = p.stdout.splitlines()
    checks = [l for l in lines if "Max Error" in l]
    perf = [l for l in lines if "AVG Performance" in l]
    assert len(checks) == 5 and all(l.rstrip().endswith("PASS)") for l in checks), checks
    assert len(perf) == 5 and all(float(l.split("=")[-1].split()[0]) > 0.1 for l in perf)
    assert f"HGEMM {layout.upper()}" in lines[0]
