"""ctypes view of include/lc_abi.h (libleetcuda_amd.so).  Plumbing only: torch is used for device memory
and streams; every compute call goes through the C-ABI.  Fails loudly if the library is missing."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "lib" / "libleetcuda_amd.so"

LC_OK, LC_ERR_ARG, LC_ERR_SHAPE, LC_ERR_HEADDIM, LC_ERR_LAUNCH, LC_ERR_VENDOR, LC_ERR_DEVICE = 0, -1, -2, -3, -4, -5, -6
LAYOUT_NN, LAYOUT_TN = 0, 1
HGEMM_AUTO, HGEMM_MFMA256, HGEMM_MFMA256P, HGEMM_GENERIC, HGEMM_MFMA256P2, HGEMM_MFMA256P3, HGEMM_MFMA128, HGEMM_MFMA256W4, HGEMM_MFMA256W4S, HGEMM_MFMA256W4B, HGEMM_MFMA256W4C = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10
ATTN_SPLIT_Q, ATTN_SHARED_QKV, ATTN_SHARED_KV, ATTN_TILING_QK, ATTN_TILING_QKV, ATTN_SPLIT_KV = range(6)

# every symbol include/lc_abi.h declares: name -> (restype, argtypes)
_vp, _i, _cp, _fp, _ip = C.c_void_p, C.c_int, C.c_char_p, C.POINTER(C.c_float), C.POINTER(C.c_int)
SYMBOLS = {
    "lc_abi_version": (_i, []),
    "lc_status_string": (_cp, [_i]),
    "lc_device_check": (_i, [_ip]),
    "lc_tune_set": (_i, [_cp, _i]),
    "lc_hgemm_f16": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "lc_vendor_init": (_i, []),
    "lc_vendor_destroy": (_i, []),
    "lc_hgemm_vendor_f16": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "lc_gemm_fp8_e4m3": (_i, [_vp, _vp, _vp, _i, _i, _i, C.c_float, _i, _vp]),
    "lc_hgemm_call": (_i, [_cp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "lc_hgemm_entry_count": (_i, []),
    "lc_hgemm_entry_name": (_cp, [_i]),
    "lc_hgemm_entry_info": (_i, [_cp, _ip, _ip]),
    "lc_attn_fwd_f16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "lc_attn_fwd_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "lc_attn_call": (_i, [_cp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "lc_attn_entry_count": (_i, []),
    "lc_attn_entry_name": (_cp, [_i]),
    "lc_attn_entry_info": (_i, [_cp, _ip, _ip, _ip, _ip, _ip, _ip]),
    "lc_hgemm_time": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _fp]),
    "lc_attn_time": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _fp]),
    "lc_probe_mfma16": (_i, [_vp, _vp, _vp, _vp]),
    "lc_probe_mfma32": (_i, [_vp, _vp, _vp, _vp]),
    "lc_probe_tr16": (_i, [_vp, _vp, _vp]),
    "lc_probe_coissue": (_i, [_i, _i, _i, _vp, _vp]),
    "lc_probe_mfma_war": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp]),
}

_lib = None


class LcError(RuntimeError):
    def __init__(self, status: int, what: str):
        self.status = status
        super().__init__(f"{what}: {status_string(status)} (lc_status {status})")


def load() -> C.CDLL:
    """dlopen the C-ABI library. `import torch` first so libamdhip64.so.7 resolves to the copy PyTorch
    already mapped (one HIP runtime per process)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m leetcuda_amd.build` "
            "(there is no CPU or PyTorch fallback for the HIP kernels)")
    try:
        import torch  # noqa: F401  (maps torch/lib/libamdhip64.so first)
        tl = Path(torch.__file__).parent / "lib" / "libhipblaslt.so"
        if tl.exists() and os.environ.get("LC_VENDOR_SYSTEM_HIPBLASLT", "0") != "1":
            C.CDLL(str(tl), mode=C.RTLD_GLOBAL)  # let dlopen("libhipblaslt.so.1") find PyTorch's copy
    except Exception:  # torch is optional for symbol-level use of the ABI
        pass
    lib = C.CDLL(str(LIB_PATH), mode=C.RTLD_GLOBAL)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def status_string(status: int) -> str:
    return load().lc_status_string(status).decode()


def check(status: int, what: str):
    if status != LC_OK:
        raise LcError(status, what)


def _ptr(t) -> int:
    return t.data_ptr()


def _stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


def _need_gpu(*tensors):
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError("leetcuda_amd: tensor must live on the MI355X (no CPU path)")
        if not t.is_contiguous():
            raise RuntimeError("leetcuda_amd: tensor must be contiguous")


def tune(key: str, value: int):
    check(load().lc_tune_set(key.encode(), value), f"lc_tune_set({key})")


def device_check() -> int:
    n = C.c_int(0)
    check(load().lc_device_check(C.byref(n)), "lc_device_check")
    return n.value


# ---- HGEMM ------------------------------------------------------------------------------------------
def hgemm(a, b, c, layout=LAYOUT_NN, variant=HGEMM_AUTO, stages=2, swizzle_stride=1):
    """c[M,N] = a[M,K] @ B ; b is [K,N] (NN) or storage [N,K] (TN). Writes c in place."""
    import torch
    _need_gpu(a, b, c)
    assert a.dtype == b.dtype == c.dtype == torch.half
    M, K = a.shape
    N = c.shape[1]
    check(load().lc_hgemm_f16(_ptr(a), _ptr(b), _ptr(c), M, N, K, layout, variant, stages, swizzle_stride,
                              _stream()), "lc_hgemm_f16")
    return c


def gemm_fp8(a8, b8_nk, c, alpha=1.0, swizzle_stride=1):
    """c[M,N] (fp16) = alpha * a8[M,K] @ b8_nk[N,K]^T; a8/b8 are torch.float8_e4m3fn (or uint8 views)."""
    _need_gpu(a8, b8_nk, c)
    M, K = a8.shape
    N = b8_nk.shape[0]
    check(load().lc_gemm_fp8_e4m3(_ptr(a8), _ptr(b8_nk), _ptr(c), M, N, K, float(alpha), swizzle_stride, _stream()),
          "lc_gemm_fp8_e4m3")
    return c


def hgemm_call(entry: str, a, b, c, stages=2, swizzle=False, swizzle_stride=1):
    _need_gpu(a, b, c)
    M, K = a.shape
    N = c.shape[1]
    check(load().lc_hgemm_call(entry.encode(), _ptr(a), _ptr(b), _ptr(c), M, N, K, stages, int(swizzle),
                               swizzle_stride, _stream()), entry)
    return c


def hgemm_vendor(a, b, c, layout=LAYOUT_NN):
    _need_gpu(a, b, c)
    M, K = a.shape
    N = c.shape[1]
    check(load().lc_hgemm_vendor_f16(_ptr(a), _ptr(b), _ptr(c), M, N, K, layout, _stream()),
          "lc_hgemm_vendor_f16")
    return c


def vendor_init():
    check(load().lc_vendor_init(), "lc_vendor_init")


def vendor_destroy():
    check(load().lc_vendor_destroy(), "lc_vendor_destroy")


def hgemm_time(a, b, c, layout=LAYOUT_NN, variant=HGEMM_AUTO, stages=2, swizzle_stride=1, warmup=2,
               iters=10) -> float:
    """Average ms per launch, HIP events on the launch stream."""
    _need_gpu(a, b, c)
    M, K = a.shape
    N = c.shape[1]
    ms = C.c_float(0)
    check(load().lc_hgemm_time(_ptr(a), _ptr(b), _ptr(c), M, N, K, layout, variant, stages, swizzle_stride,
                               warmup, iters, _stream(), C.byref(ms)), "lc_hgemm_time")
    return ms.value


def hgemm_entries():
    lib = load()
    out = []
    for i in range(lib.lc_hgemm_entry_count()):
        name = lib.lc_hgemm_entry_name(i)
        lay, na = C.c_int(), C.c_int()
        lib.lc_hgemm_entry_info(name, C.byref(lay), C.byref(na))
        out.append((name.decode(), lay.value, na.value))
    return out


# ---- attention --------------------------------------------------------------------------------------
def attn_fwd(q, k, v, o, v_transposed=False, family=ATTN_SPLIT_Q, acc_f32=False, stages=2):
    import torch
    _need_gpu(q, k, v, o)
    assert q.dtype == k.dtype == v.dtype == o.dtype == torch.half
    B, H, N, D = q.shape
    check(load().lc_attn_fwd_f16(_ptr(q), _ptr(k), _ptr(v), _ptr(o), B, H, N, D, int(v_transposed), family,
                                 int(acc_f32), stages, _stream()), "lc_attn_fwd_f16")
    return o


def attn_fwd_bf16(q, k, v, o):
    """bfloat16 forward for D in {256, 512} (BASELINE config 5 extension)."""
    import torch
    _need_gpu(q, k, v, o)
    assert q.dtype == k.dtype == v.dtype == o.dtype == torch.bfloat16
    B, H, N, D = q.shape
    check(load().lc_attn_fwd_bf16(_ptr(q), _ptr(k), _ptr(v), _ptr(o), B, H, N, D, _stream()), "lc_attn_fwd_bf16")
    return o


def attn_call(entry: str, q, k, v, o, stages=2):
    _need_gpu(q, k, v, o)
    B, H, N, D = q.shape
    check(load().lc_attn_call(entry.encode(), _ptr(q), _ptr(k), _ptr(v), _ptr(o), B, H, N, D, stages,
                              _stream()), entry)
    return o


def attn_time(q, k, v, o, v_transposed=False, family=ATTN_SPLIT_Q, stages=2, warmup=1, iters=5) -> float:
    _need_gpu(q, k, v, o)
    B, H, N, D = q.shape
    ms = C.c_float(0)
    check(load().lc_attn_time(_ptr(q), _ptr(k), _ptr(v), _ptr(o), B, H, N, D, int(v_transposed), family,
                              stages, warmup, iters, _stream(), C.byref(ms)), "lc_attn_time")
    return ms.value


def attn_entries():
    lib = load()
    out = []
    for i in range(lib.lc_attn_entry_count()):
        name = lib.lc_attn_entry_name(i)
        v = [C.c_int() for _ in range(6)]
        lib.lc_attn_entry_info(name, *[C.byref(x) for x in v])
        out.append((name.decode(),) + tuple(x.value for x in v))
    return out
