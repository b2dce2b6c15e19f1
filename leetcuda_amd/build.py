"""Build every native artefact in-tree (no JIT cache): the gfx950 C-ABI library, the C oracle and the
two PyTorch extension modules that re-export the reference's entry points.

    python -m leetcuda_amd.build [--force] [--no-torch-ext]

hipcc cross-compiles gfx950 without a GPU, so this runs in the CPU-only dev container as well.
"""
from __future__ import annotations

import argparse
import json
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


def _digest(paths, extra="") -> str:
    """sha256 over the CONTENTS of `paths` (+ `extra`): what decides whether a built artefact is current.  mtimes do not survive
    the snapshot that carries the tree to a GPU box (every box used to rebuild the library for ~140 s before its first test)."""
    import hashlib
    h = hashlib.sha256(extra.encode())
    for q in sorted(Path(x) for x in paths):
        h.update(str(q.relative_to(ROOT) if q.is_relative_to(ROOT) else q).encode())
        h.update(q.read_bytes() if q.exists() else b"<missing>")
    return h.hexdigest()


def _stamps() -> dict:
    try:
        return json.loads((LIBDIR / "stamps.json").read_text())
    except Exception:
        return {}


def _fresh(target: Path, key: str, digest: str) -> bool:
    return target.exists() and _stamps().get(key) == digest


def _mark(key: str, digest: str):
    LIBDIR.mkdir(parents=True, exist_ok=True)
    st = _stamps()
    st[key] = digest
    (LIBDIR / "stamps.json").write_text(json.dumps(st, indent=1))


def _run(cmd, **kw):
    print("[build]", " ".join(str(c) for c in cmd), flush=True)
    subprocess.run([str(c) for c in cmd], check=True, **kw)


def hipcc() -> str:
    exe = shutil.which("hipcc") or str(ROCM / "bin" / "hipcc")
    if not Path(exe).exists():
        raise RuntimeError("hipcc not found: the MI355X kernels cannot be built")
    return exe


def _flags():
    diag = os.environ.get("LC_DIAG") == "1"
    return ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-fno-honor-nans",
            "-mno-amdgpu-ieee", *(["-DLC_DIAG"] if diag else []), f"-I{ROOT / 'include'}"]


def _portable(flags):
    """flags with the checkout's absolute path replaced: the GPU box runs the same tree from another directory"""
    return [str(f).replace(str(ROOT), "$ROOT") for f in flags]


def _stamp_ok(stamp: Path, flags) -> bool:
    """The .so is only current if it was built with THESE flags (toggling LC_DIAG must rebuild: a diagnosis library
    cannot be told from a production one by mtimes)."""
    try:
        return json.loads(stamp.read_text())["flags"] == _portable(flags)
    except Exception:
        return False


def _unit_deps(dep: Path):
    txt = dep.read_text().replace("\\\n", " ")
    return [Path(d) for d in (txt.split(":", 1)[1].split() if ":" in txt else [])]


def _unit_digest(dep: Path, flags) -> str:
    """Content digest of everything the compiler read for one translation unit (its -MD dependency list; system headers
    included) + the flags."""
    import hashlib
    h = hashlib.sha256(" ".join(_portable(flags)).encode())
    for q in sorted(_unit_deps(dep), key=lambda x: (x.name, str(x))):
        h.update(q.name.encode())        # names, not paths: the tree sits at another path on a GPU box
        h.update(q.read_bytes() if q.exists() else b"<missing>")
    return h.hexdigest()


def _unit_current(obj: Path, asm, dep: Path, flags) -> bool:
    """A unit is current when its object (and audit assembly) exist and the CONTENT of every file in its dependency list is
    what it was when the object was built (stamps.json "unit:<stem>").  Not mtimes: they do not survive the snapshot to a GPU
    box, and a stale object re-linked under a fresh library digest would never be rebuilt again (round-3 advisor finding)."""
    if not obj.exists() or not dep.exists() or (asm is not None and not asm.exists()):
        return False
    deps = _unit_deps(dep)
    if not deps or not all(d.exists() for d in deps):
        return False
    have = _stamps().get("unit:" + obj.stem)
    if have is None:   # no recorded digest (a tree built before the per-unit stamps existed, or a lost stamps.json): rebuild once — mtimes
        return False   # say nothing after a snapshot, and a stale object accepted here would be stamped as current for good (round-4 advisor)
    return have == _unit_digest(dep, flags)


def build_abi(force: bool = False, audit: bool = True) -> Path:
    """hipcc --offload-arch=gfx950 -> leetcuda_amd/lib/libleetcuda_amd.so (the C-ABI, include/lc_abi.h), then the
    ISA audit of leetcuda_amd/isa_audit.py on the device assembly of every translation unit (same flags)."""
    LIBDIR.mkdir(parents=True, exist_ok=True)
    out = LIBDIR / LIB_NAME
    stamp = LIBDIR / "build_stamp.json"
    srcs = sorted(CSRC.glob("*.hip")) + sorted(CSRC.glob("*.h")) + sorted(CSRC.glob("*.inc"))
    srcs.append(ROOT / "include" / "lc_abi.h")
    srcs.append(PKG / "isa_audit.py")
    srcs.append(ROOT / "tools" / "gen_hgemm_w4y.py")
    srcs.append(ROOT / "tools" / "gen_attn_w4i.py")
    srcs.append(ROOT / "tools" / "gen_gemm_fp8_w4k.py")
    srcs.append(ROOT / "tools" / "gen_attn_bigd7.py")
    flags = _flags()
    # the generated K loops (hgemm_w4y_loop*.inc) must be what tools/gen_hgemm_w4y.py emits today
    gen = subprocess.run([sys.executable, str(ROOT / "tools" / "gen_hgemm_w4y.py"), "--check"], capture_output=True, text=True)
    if gen.returncode != 0:
        raise RuntimeError("generated sources are stale: " + gen.stderr.strip())
    for script in ("gen_attn_w4i.py", "gen_gemm_fp8_w4k.py", "gen_attn_bigd7.py"):
        gen = subprocess.run([sys.executable, str(ROOT / "tools" / script), "--check"], capture_output=True, text=True)
        if gen.returncode != 0:
            raise RuntimeError("generated sources are stale: " + gen.stderr.strip())
    if os.environ.get("LC_DIAG") == "1":   # the ablation loops (results WRONG by design) exist only inside a diagnosis build
        subprocess.run([sys.executable, str(ROOT / "tools" / "gen_hgemm_w4y.py"), "--diag", str(LIBDIR / "gen")], check=True)
        flags = flags + [f"-I{LIBDIR / 'gen'}"]
    dg = _digest(srcs, " ".join(_portable(flags)))
    if not force and _fresh(out, "abi", dg) and _stamp_ok(stamp, flags):
        return out
    # translation units (lc_abi.hip + the compile-heavy literal-AGPR kernels in tu_*.hip), compiled in parallel; each
    # is compiled twice: to an object and (device side only) to assembly for the audit
    objdir = LIBDIR / "obj"
    objdir.mkdir(exist_ok=True)
    units = [CSRC / "lc_abi.hip"] + sorted(CSRC.glob("tu_*.hip"))
    for old in list(objdir.glob("tu_*.*")):      # objects / assembly of translation units that no longer exist
        if old.suffix in (".o", ".s", ".d") and not (CSRC / (old.stem + ".hip")).exists():
            old.unlink()
    flags_ok = _stamp_ok(stamp, flags)
    procs = []
    for u in units:
        obj = objdir / (u.stem + ".o")
        asm = objdir / (u.stem + ".s")
        dep = objdir / (u.stem + ".d")
        # per-unit staleness from the compiler's own dependency file (-MD): only units whose sources changed rebuild
        if not force and flags_ok and _unit_current(obj, asm if audit else None, dep, flags):
            procs.append((u, obj, None))
            continue
        for stale in (obj, asm):         # never leave yesterday's object / assembly behind a failed compile
            if stale.exists():
                stale.unlink()
        cmd = [hipcc(), *flags, "-MD", "-MF", str(dep), "-c", "-o", str(obj), str(u)]
        print("[build] " + " ".join(cmd), flush=True)
        procs.append((u, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        if audit:
            cmd = [hipcc(), *flags, "-S", "--cuda-device-only", "-Wno-unused-command-line-argument", "-o", str(asm), str(u)]
            procs.append((u, None, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    objs = []
    for u, obj, pr in procs:
        if pr is not None:
            log, _ = pr.communicate()
            if pr.returncode != 0:
                sys.stdout.write(log[-8000:])
                raise subprocess.CalledProcessError(pr.returncode, pr.args)
        if obj is not None:
            objs.append(obj)
            if pr is not None:     # rebuilt: remember what it was built from
                _mark("unit:" + obj.stem, _unit_digest(objdir / (obj.stem + ".d"), flags))
    if audit:
        from leetcuda_amd import isa_audit
        reps, bad = isa_audit.audit_files([objdir / (u.stem + ".s") for u in units])
        for r in reps:
            print(f"[audit] {r.name}: vgpr {r.vgpr_count} agpr {r.agpr_count} scratch {r.scratch} "
                  f"asm loads {r.asm_loads} compiler v_accvgpr {r.compiler_accvgpr}", flush=True)
        (objdir / "isa_audit.json").write_text(json.dumps(
            [{"kernel": r.name, "vgpr": r.vgpr_count, "agpr": r.agpr_count, "scratch": r.scratch,
              "asm_loads": r.asm_loads, "compiler_accvgpr": r.compiler_accvgpr, "violations": r.violations}
             for r in reps], indent=1))
        if bad:
            if out.exists():
                out.unlink()     # never leave a library around whose hidden-state invariants do not hold
            raise RuntimeError("ISA audit failed (leetcuda_amd/isa_audit.py):\n  " + "\n  ".join(bad))
    _run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs, "-ldl"])
    stamp.write_text(json.dumps({"flags": _portable(flags)}))
    _mark("abi", dg)
    return out


def build_diag(force: bool = False) -> Path:
    """liblc_diag.so: the hardware probes of include/lc_diag.h + the ablated copies of the generated attention stream (tests /
    tools only; not part of the drop-in library).  The ablated phase statements are generated into lib/gen/ (never tracked)."""
    LIBDIR.mkdir(parents=True, exist_ok=True)
    out = LIBDIR / "liblc_diag.so"
    srcs = sorted((CSRC / "diag").glob("*.hip")) + [CSRC / "lc_common.h", ROOT / "include" / "lc_diag.h", CSRC / "attn_w4i.hip", CSRC / "attn_w4u.hip", CSRC / "attn_mp.h", CSRC / "attn_fwd.hip",
                                                     CSRC / "hgemm_mid.hip", ROOT / "tools" / "gen_attn_w4i.py"]
    dg = _digest(srcs)
    if not force and _fresh(out, "diag", dg):
        return out
    gen = LIBDIR / "gen"
    subprocess.run([sys.executable, str(ROOT / "tools" / "gen_attn_w4i.py"), "--diag", str(gen)], check=True)
    sys.path.insert(0, str(ROOT / "tools"))
    try:
        import gen_attn_w4i
        abls = gen_attn_w4i.ABLATIONS
    finally:
        sys.path.pop(0)
    objdir = LIBDIR / "obj"
    objdir.mkdir(exist_ok=True)
    base = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-inline-asm",
            "-fno-honor-nans", "-mno-amdgpu-ieee", f"-I{ROOT / 'include'}", f"-I{gen}", f"-I{CSRC}"]
    jobs = [(objdir / "diag_main.o", [*base, "-c", "-o", objdir / "diag_main.o", CSRC / "diag" / "lc_diag.hip"]),
            (objdir / "diag_w4u_stamps.o", [*base, "-c", "-o", objdir / "diag_w4u_stamps.o", CSRC / "diag" / "attn_w4u_stamps.hip"]),
            (objdir / "diag_mid256.o", [*base, "-c", "-o", objdir / "diag_mid256.o", CSRC / "diag" / "mid256_probe.hip"])]
    for k in abls:
        o = objdir / f"diag_w4i_abl{k}.o"
        jobs.append((o, [*base, f"-DW4I_ABL={k}", f'-DW4I_INC32="attn_w4i_d32_abl{k}.inc"', f'-DW4I_INC64="attn_w4i_d64_abl{k}.inc"', f'-DW4I_INC96="attn_w4i_d96_abl{k}.inc"', f'-DW4I_INC128="attn_w4i_d128_abl{k}.inc"',
                         "-c", "-o", o, CSRC / "diag" / "attn_w4i_abl.hip"]))
    procs = []
    for o, cmd in jobs:
        print("[build]", " ".join(str(c) for c in cmd), flush=True)
        procs.append((o, subprocess.Popen([str(c) for c in cmd], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for o, pr in procs:
        log, _ = pr.communicate()
        if pr.returncode != 0:
            sys.stdout.write(log[-6000:])
            raise subprocess.CalledProcessError(pr.returncode, pr.args)
    _run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *[o for o, _ in jobs]])
    _mark("diag", dg)
    return out


def build_cpp_bench(force: bool = False) -> Path:
    """tools/cpp/hgemm_bench.bin: the torch-free C++ bench / error-check harness; links only the C-ABI."""
    d = ROOT / "tools" / "cpp"
    out = d / "hgemm_bench.bin"
    build_abi(False)
    dg = _digest([d / "hgemm_bench.cpp", d / "Makefile", ROOT / "include" / "lc_abi.h"])   # (links the C-ABI dynamically: the .so's contents do not matter)
    if not force and _fresh(out, "cpp_bench", dg):
        return out
    _run(["make", "-C", d, "-B"])
    _mark("cpp_bench", dg)
    return out


def build_oracle(force: bool = False) -> Path:
    """gcc -> oracle/liblc_oracle.so (CPU restatement of the reference algorithms; TEST INFRASTRUCTURE)."""
    out = ORACLE / ORACLE_NAME
    srcs = [ORACLE / "lc_oracle.c", ORACLE / "lc_oracle.h"]
    dg = _digest(srcs)
    if not force and _fresh(out, "oracle", dg):
        return out
    cc = shutil.which("gcc") or "cc"
    _run([cc, "-O3", "-mavx2", "-mfma", "-fopenmp", "-fPIC", "-shared", "-std=c11", "-o", out,
          ORACLE / "lc_oracle.c", "-lm"])
    _mark("oracle", dg)
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
        dg = _digest([src, CSRC / "torch" / "torch_shim.h", ROOT / "include" / "lc_abi.h"], extra="rpath: $ORIGIN/lib, $ORIGIN/leetcuda_amd/lib")
        if not force and _fresh(out, f"torch_{name}", dg):
            outs.append(out)
            continue
        cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1",
               "-DUSE_ROCM=1", f"-DTORCH_EXTENSION_NAME={name}", "-DTORCH_API_INCLUDE_EXTENSION_H",
               f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", *inc, src, "-o", out,
               f"-L{LIBDIR}", "-lleetcuda_amd", "-Wl,-rpath,$ORIGIN/lib", "-Wl,-rpath,$ORIGIN/leetcuda_amd/lib", f"-L{torch_lib}",   # (in-tree / installed top-level: setup.py)
               "-lc10", "-lc10_hip", "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-ltorch_python",
               f"-Wl,-rpath,{torch_lib}"]
        _run(cmd)
        _mark(f"torch_{name}", dg)
        outs.append(out)
    return outs


def build_all(force: bool = False, torch_ext: bool = True):
    abi = build_abi(force)
    build_diag(force)
    build_cpp_bench(force)
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
