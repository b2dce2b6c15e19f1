#!/usr/bin/env python3
"""Run the reference's OWN bench scripts, unmodified, on top of this library.

    python tools/run_reference_bench.py /path/to/LeetCUDA/kernels/hgemm/hgemm.py        [--M 8192 --N 8192 --K 8192 --mma-all ...]
    python tools/run_reference_bench.py /path/to/LeetCUDA/kernels/flash-attn/flash_attn_mma.py [--B 4 --H 32 --N 4096 --D 128 ...]

north_star: "keeping the same PyTorch extension entry points so the existing Python benches drop in unchanged".  The
scripts are executed with runpy from their own directory; nothing in them is edited.  Three things stand between an
untouched script and an MI355X box, and all three are satisfied from OUTSIDE the script:

  1. `torch.utils.cpp_extension.load(name="hgemm_lib" | "flash_attn_lib", sources=[...*.cu])`
     (kernels/hgemm/tools/utils.py:110-131, kernels/flash-attn/flash_attn_mma.py:222-228) would nvcc-compile the CUDA
     sources.  It is replaced by a function that returns the prebuilt drop-in module of the same export list
     (leetcuda_amd/toy_hgemm*.so, leetcuda_amd/flash_attn_lib*.so: pybind modules over the C-ABI).
  2. `import toy_hgemm` (tools/utils.py:131) resolves to leetcuda_amd/toy_hgemm*.so through sys.path.
  3. `from flash_attn import flash_attn_func` (flash_attn_mma.py:11): the pip package `flash-attn` is the reference's
     COMPARATOR, absent here; a stub module provides flash_attn_func([B,N,H,D]) on torch's own fused SDPA.

LC_SHIM_DRYRUN=1 (CPU containers, tests/test_reference_shim.py): torch.cuda device-name queries are answered without
a device, so that a GPU-less run can prove that the interception points are reached with zero source edits; the script
then stops at its first device="cuda" allocation.
"""
from __future__ import annotations

import os
import runpy
import sys
import types
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
PKG = ROOT / "leetcuda_amd"
MODULE_FOR = {"hgemm_lib": "toy_hgemm", "toy_hgemm": "toy_hgemm", "flash_attn_lib": "flash_attn_lib"}


def log(msg: str):
    print(f"[lc-shim] {msg}", flush=True)


def install(dryrun: bool = False):
    """Install the three interceptions into this interpreter (idempotent)."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.utils.cpp_extension as cpp_ext

    if str(PKG) not in sys.path:
        sys.path.append(str(PKG))             # `import toy_hgemm` / `import flash_attn_lib`

    def load(name, sources=None, *args, **kwargs):
        mod = MODULE_FOR.get(name)
        if mod is None:
            raise RuntimeError(f"[lc-shim] cpp_extension.load(name={name!r}): no MI355X drop-in module of that name")
        n_cu = len([s for s in (sources or []) if str(s).endswith((".cu", ".cc"))])
        log(f"cpp_extension.load(name={name!r}, {n_cu} CUDA sources) -> prebuilt {PKG.name}/{mod} (no nvcc, no CUDA)")
        return __import__(mod)

    cpp_ext.load = load
    torch.utils.cpp_extension.load = load

    if "flash_attn" not in sys.modules:
        fa = types.ModuleType("flash_attn")
        fa.__doc__ = "stub of the pip package flash-attn (the reference's comparator), backed by torch SDPA"

        def flash_attn_func(q, k, v, dropout_p=0.0, softmax_scale=None, causal=False, **_):
            # flash-attn layout [B, N, H, D] (flash_attn_mma.py:1115 transposes before the call)
            import torch.nn.functional as F
            o = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2),
                                               dropout_p=dropout_p, is_causal=causal, scale=softmax_scale)
            return o.transpose(1, 2)

        fa.flash_attn_func = flash_attn_func
        fa.__version__ = "0.0.0+lc-shim-sdpa"
        sys.modules["flash_attn"] = fa
        log("sys.modules['flash_attn'] = stub (flash_attn_func on torch SDPA)")

    if dryrun and not torch.cuda.is_available():
        torch.cuda.current_device = lambda: 0
        torch.cuda.get_device_name = lambda *_a, **_k: "AMD Instinct MI355X (dry run: no device in this container)"
        torch.cuda.get_device_capability = lambda *_a, **_k: (9, 5)
        torch.cuda.manual_seed_all = lambda *_a, **_k: None
        log("dry run: torch.cuda device-name queries answered without a device")


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if not argv or argv[0] in ("-h", "--help"):
        print(__doc__)
        return 2
    script = Path(argv[0]).resolve()
    if not script.is_file():
        raise SystemExit(f"[lc-shim] no such script: {script}")
    install(dryrun=os.environ.get("LC_SHIM_DRYRUN") == "1")
    os.chdir(script.parent)                    # the scripts use ./relative source paths and `from tools.utils import`
    sys.path.insert(0, str(script.parent))
    sys.argv = [str(script)] + list(argv[1:])
    log(f"runpy {script} {' '.join(argv[1:])}  (cwd {script.parent}; source untouched)")
    runpy.run_path(str(script), run_name="__main__")
    return 0


if __name__ == "__main__":
    sys.exit(main())
