"""Build every native artefact in-tree (no JIT cache): the gfx950 C-ABI library, the C oracle and the
two PyTorch extension modules that re-export the reference's entry points.

    python -m leetcuda_amd.build [--force] [--no-torch-ext]

hipcc cross-compiles gfx950 without a GPU, so this runs in the CPU-only dev container as well.
"""
from __future__ import annotations

import argparse
import os
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
PKG = ROOT / "leetcuda_amd"
CSRC = PKG / "csrc"
LIBDIR = PKG / "lib"
ORACLE = ROOT / "oracle"
ROCM = Path(os.environ.get("ROCM_PATH", "/opt/rocm"))

LIB_NAME = "libleetcuda_amd.so"
ORACLE_NAME = "liblc_oracle.so"


def _newer(target: Path, sources) -> bool:
    """True when `target` exists and is newer than every source."""
    if not target.exists():
        return False
    t = target.stat().st_mtime
    return all(Path(s).stat().st_mtime <= t for s in sources)


def _run(cmd, **kw):
    print("[build]", " ".join(str(c) for c in cmd), flush=True)
    subprocess.run([str(c) for c in cmd], check=True, **kw)


def hipcc() -> str:
    exe = shutil.which("hipcc") or str(ROCM / "bin" / "hipcc")
    if not Path(exe).exists():
        raise RuntimeError("hipcc not found: the MI355X kernels cannot be built")
    return exe


def build_abi(force: bool = False) -> Path:
    """hipcc --offload-arch=gfx950 -> leetcuda_amd/lib/libleetcuda_amd.so (the C-ABI, include/lc_abi.h)."""
    LIBDIR.mkdir(parents=True, exist_ok=True)
    out = LIBDIR / LIB_NAME
    srcs = sorted(CSRC.glob("*.hip")) + sorted(CSRC.glob("*.h")) + sorted(CSRC.glob("*.inc"))
    srcs.append(ROOT / "include" / "lc_abi.h")
    if not force and _newer(out, srcs):
        return out
    # four translation units (lc_abi.hip + the compile-heavy literal-AGPR kernels in tu_*.hip), compiled in parallel
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-fno-honor-nans",
             "-mno-amdgpu-ieee", *(["-DLC_DIAG"] if os.environ.get("LC_DIAG") == "1" else []), f"-I{ROOT / 'include'}"]
    objdir = LIBDIR / "obj"
    objdir.mkdir(exist_ok=True)
    units = [CSRC / "lc_abi.hip"] + sorted(CSRC.glob("tu_*.hip"))
    procs = []
    for u in units:
        obj = objdir / (u.stem + ".o")
        cmd = [hipcc(), *flags, "-c", "-o", str(obj), str(u)]
        print("[build] " + " ".join(cmd), flush=True)
        procs.append((u, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    objs = []
    for u, obj, pr in procs:
        log, _ = pr.communicate()
        if pr.returncode != 0:
            sys.stdout.write(log)
            raise subprocess.CalledProcessError(pr.returncode, pr.args)
        objs.append(obj)
    _run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs, "-ldl"])
    return out


def build_oracle(force: bool = False) -> Path:
    """gcc -> oracle/liblc_oracle.so (CPU restatement of the reference algorithms; TEST INFRASTRUCTURE)."""
    out = ORACLE / ORACLE_NAME
    srcs = [ORACLE / "lc_oracle.c", ORACLE / "lc_oracle.h"]
    if not force and _newer(out, srcs):
        return out
    cc = shutil.which("gcc") or "cc"
    _run([cc, "-O3", "-mavx2", "-mfma", "-fopenmp", "-fPIC", "-shared", "-std=c11", "-o", out,
          ORACLE / "lc_oracle.c", "-lm"])
    return out


def build_torch_ext(force: bool = False):
    """The drop-in modules `toy_hgemm` and `flash_attn_lib` (pybind11 + torch/extension.h, C++ only:
    they carry no device code, they forward to the C-ABI). Built in-tree next to the package."""
    import sysconfig

    import torch
    from torch.utils import cpp_extension

    suffix = sysconfig.get_config_var("EXT_SUFFIX") or ".so"
    outs = []
    abi = build_abi(force=False)
    inc = [f"-I{p}" for p in cpp_extension.include_paths()] + [f"-I{ROOT / 'include'}"]
    inc += [f"-I{sysconfig.get_paths()['include']}", f"-I{ROCM / 'include'}"]
    torch_lib = Path(torch.__file__).parent / "lib"
    cxx = shutil.which("g++") or "c++"
    for name in ("toy_hgemm", "flash_attn_lib"):
        src = CSRC / "torch" / f"{name}.cpp"
        out = PKG / f"{name}{suffix}"
        if not force and _newer(out, [src, CSRC / "torch" / "torch_shim.h", abi]):
            outs.append(out)
            continue
        cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1",
               "-DUSE_ROCM=1", f"-DTORCH_EXTENSION_NAME={name}", "-DTORCH_API_INCLUDE_EXTENSION_H",
               f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", *inc, src, "-o", out,
               f"-L{LIBDIR}", "-lleetcuda_amd", f"-Wl,-rpath,$ORIGIN/lib", f"-L{torch_lib}",
               "-lc10", "-lc10_hip", "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-ltorch_python",
               f"-Wl,-rpath,{torch_lib}"]
        _run(cmd)
        outs.append(out)
    return outs


def build_all(force: bool = False, torch_ext: bool = True):
    abi = build_abi(force)
    orc = build_oracle(force)
    ext = build_torch_ext(force) if torch_ext else []
    return abi, orc, ext


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--no-torch-ext", action="store_true")
    a = ap.parse_args()
    res = build_all(a.force, not a.no_torch_ext)
    print("[build] ok:", res)
    sys.exit(0)
