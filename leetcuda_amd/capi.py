"""ctypes view of include/lc_abi.h (libleetcuda_amd.so).  Plumbing only: torch is used for device memory
and streams; every compute call goes through the C-ABI.  Fails loudly if the library is missing."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "lib" / "libleetcuda_amd.so"
DIAG_LIB_PATH = _PKG / "lib" / "liblc_diag.so"   # probes of include/lc_diag.h (tests / tools only)

LC_OK, LC_ERR_ARG, LC_ERR_SHAPE, LC_ERR_HEADDIM, LC_ERR_LAUNCH, LC_ERR_VENDOR, LC_ERR_DEVICE = 0, -1, -2, -3, -4, -5, -6
LAYOUT_NN, LAYOUT_TN = 0, 1
# lc_hgemm_variant (values 2, 5, 7, 8 were retired round-1 experiments: LC_ERR_ARG)
HGEMM_AUTO, HGEMM_MFMA256, HGEMM_GENERIC, HGEMM_MFMA256P2, HGEMM_MFMA128 = 0, 1, 3, 4, 6
HGEMM_VALU_NAIVE, HGEMM_VALU_SLICED_K, HGEMM_VALU_T8X8_X4, HGEMM_VALU_T16X8_K32 = 20, 21, 22, 30   # the VALU ladder: 20..30
HGEMM_MFMA256W4B, HGEMM_MFMA256W4C, HGEMM_MFMA256W4X, HGEMM_MFMA256W4Y = 9, 10, 12, 13
HGEMM_MID = 14   # the one-round kernel (hgemm_mid.hip)
HGEMM_EDGE = 15  # the vectorised edge kernel (hgemm_edge.hip): any M, N; K % 8 == 0 (NN: N % 8 == 0)
HGEMM_RAGGED = 16  # ragged M / N, K % 32 == 0, N % 8 == 0: the tiled kernels, clamped 128 x 128 tiles of the mid-size kernel on what they do not divide
HGEMM_KPAD = 17    # K % 32 != 0 on a large problem: zero-padded operand copies in the workspace + the tuned kernels
ATTN_SPLIT_Q, ATTN_SHARED_QKV, ATTN_SHARED_KV, ATTN_TILING_QK, ATTN_TILING_QKV, ATTN_SPLIT_KV = range(6)

# every symbol include/lc_abi.h declares: name -> (restype, argtypes)
_vp, _i, _cp, _fp, _ip = C.c_void_p, C.c_int, C.c_char_p, C.POINTER(C.c_float), C.POINTER(C.c_int)
SYMBOLS = {
    "lc_abi_version": (_i, []),
    "lc_status_string": (_cp, [_i]),
    "lc_device_check": (_i, [_ip]),
    "lc_build_info": (_cp, [_ip]),
    "lc_tune_set": (_i, [_cp, _i]),
    "lc_tune_get": (_i, [_cp, _ip, _ip]),
    "lc_tune_count": (_i, []),
    "lc_tune_key": (_cp, [_i]),
    "lc_tune_calibrate": (_i, [_vp, _fp]),
    "lc_workspace_release": (C.c_size_t, []),
    "lc_workspace_bytes": (C.c_size_t, []),
    "lc_hgemm_f16": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "lc_vendor_init": (_i, []),
    "lc_vendor_destroy": (_i, []),
    "lc_hgemm_vendor_f16": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "lc_gemm_fp8_e4m3": (_i, [_vp, _vp, _vp, _i, _i, _i, C.c_float, _i, _vp]),
    "lc_mxfp8_pack_scales": (_i, [_vp, _vp, _i, _i, _vp]),
    "lc_gemm_mxfp8": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, C.c_float, _i, _vp]),
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
    "lc_hgemm_kernel_name": (_i, [_i, _i, _i, _i, _i, _cp, _i]),
    "lc_attn_kernel_name": (_i, [_i, _i, _i, _i, _cp, _i]),
    "lc_attn_kernel_name_bh": (_i, [_i, _i, _i, _i, _i, _cp, _i]),
    "lc_attn_slowpath_stats": (_i, [C.POINTER(C.c_uint), _i]),
    "lc_timer_start": (_i, [_vp, C.POINTER(_vp)]),
    "lc_timer_stop": (_i, [_vp, _fp]),
    "lc_clock_probe": (_i, [_vp, _vp]),
}
# every symbol include/lc_diag.h declares (liblc_diag.so)
DIAG_SYMBOLS = {
    "lc_probe_mid256": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "lc_probe_mfma16": (_i, [_vp, _vp, _vp, _vp]),
    "lc_probe_mfma32": (_i, [_vp, _vp, _vp, _vp]),
    "lc_probe_tr16": (_i, [_vp, _vp, _vp]),
    "lc_probe_coissue": (_i, [_i, _i, _i, _vp, _vp]),
    "lc_probe_attn_mix": (_i, [_i, _i, _vp, _vp]),
    "lc_probe_mfma_form": (_i, [_i, _vp, _vp]),
    "lc_probe_mfma_war": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp]),
    "lc_diag_pollute": (_i, [C.c_uint, _i, _vp]),
    "lc_diag_attn_w4i": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "lc_diag_attn_w4u_stamps": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
}

_lib = None
_diag = None


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


def load_diag() -> C.CDLL:
    """dlopen liblc_diag.so (hardware probes; tests and tools only)."""
    global _diag
    if _diag is not None:
        return _diag
    if not DIAG_LIB_PATH.exists():
        raise RuntimeError(f"{DIAG_LIB_PATH} is missing: build it with `python -m leetcuda_amd.build`")
    import torch  # noqa: F401  (maps libamdhip64 first)
    lib = C.CDLL(str(DIAG_LIB_PATH), mode=C.RTLD_GLOBAL)
    for name, (res, args) in DIAG_SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _diag = lib
    return lib


def build_info():
    """(flags string, is_diag) of the loaded library."""
    d = C.c_int(0)
    return load().lc_build_info(C.byref(d)).decode(), bool(d.value)


def require_production():
    """bench.py and the parity tests refuse a LC_DIAG=1 library: its diagnosis knobs can make results WRONG."""
    info, diag = build_info()
    if diag:
        raise RuntimeError(f"libleetcuda_amd.so is a DIAGNOSIS build ({info}); rebuild with "
                           "`python -m leetcuda_amd.build --force` (LC_DIAG unset)")


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


def _shape_err(what: str):
    # the reference wrappers' message (hgemm_mma_stage.cu:2048-2057 CHECK_TORCH_TENSOR_SHAPE)
    raise RuntimeError(f"Tensor size mismatch! ({what})")


def _gemm_dims(a, b, c, layout=LAYOUT_NN, b_is_nk=False):
    """(M, N, K) after checking a [M,K], b and c [M,N] — a mismatched tensor must fail here with the reference's
    message, not as an out-of-bounds access on the device.  b: NN -> [K,N]; TN -> the reference presents a
    [K,N]-shaped tensor whose STORAGE is [N,K] (tools/utils.py:152-156), so either orientation is accepted;
    b_is_nk (fp8 entry) -> [N,K]."""
    if a.dim() != 2 or b.dim() != 2 or c.dim() != 2:
        _shape_err("2-D tensors expected")
    M, K = a.shape
    N = c.shape[1]
    if c.shape[0] != M:
        _shape_err(f"c {tuple(c.shape)} vs a {tuple(a.shape)}")
    if b_is_nk:
        ok = tuple(b.shape) == (N, K)
    elif layout == LAYOUT_TN:
        ok = tuple(b.shape) in ((K, N), (N, K))
    else:
        ok = tuple(b.shape) == (K, N)
    if not ok:
        _shape_err(f"b {tuple(b.shape)} vs K={K}, N={N}")
    return M, N, K


def _attn_dims(q, k, v, o, v_transposed=False):
    if q.dim() != 4:
        _shape_err("4-D [B,H,N,D] tensors expected")
    B, H, N, D = q.shape
    vshape = (B, H, D, N) if v_transposed else (B, H, N, D)
    if tuple(k.shape) != (B, H, N, D) or tuple(o.shape) != (B, H, N, D) or tuple(v.shape) != vshape:
        _shape_err(f"q {tuple(q.shape)} k {tuple(k.shape)} v {tuple(v.shape)} o {tuple(o.shape)}")
    return B, H, N, D


def _need_gpu(*tensors):
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError("leetcuda_amd: tensor must live on the MI355X (no CPU path)")
        if not t.is_contiguous():
            raise RuntimeError("leetcuda_amd: tensor must be contiguous")


def tune(key: str, value: int):
    check(load().lc_tune_set(key.encode(), value), f"lc_tune_set({key})")


def tune_get(key: str):
    """(current, default) of a knob."""
    v, d = C.c_int(0), C.c_int(0)
    check(load().lc_tune_get(key.encode(), C.byref(v), C.byref(d)), f"lc_tune_get({key})")
    return v.value, d.value


def tune_items():
    """{key: (current, default)} of every knob this library accepts."""
    lib = load()
    out = {}
    for i in range(lib.lc_tune_count()):
        k = lib.lc_tune_key(i)
        if k:
            out[k.decode()] = tune_get(k.decode())
    return out


def tune_calibrate():
    """Measure the split-KV rule's constants on the current device (lc_tune_calibrate).  Returns {"adopted": bool, "tau128_us", "tau64_us",
    "x0_us", "bytes_per_us"}; adopted False = the measurement was refused as implausible and the built-in constants stay."""
    out = (C.c_float * 4)()
    rc = load().lc_tune_calibrate(_stream(), out)
    if rc not in (LC_OK, LC_ERR_ARG):
        check(rc, "lc_tune_calibrate")
    return {"adopted": rc == LC_OK, "tau128_us": out[0], "tau64_us": out[1], "x0_us": out[2], "bytes_per_us": out[3]}


def workspace_bytes() -> int:
    """Bytes of internal workspace (split-KV / split-K partials) currently cached, all devices."""
    return int(load().lc_workspace_bytes())


def workspace_release() -> int:
    """hipFree every cached workspace buffer (waits for the devices); returns the bytes given back."""
    return int(load().lc_workspace_release())


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
    M, N, K = _gemm_dims(a, b, c, layout)
    check(load().lc_hgemm_f16(_ptr(a), _ptr(b), _ptr(c), M, N, K, layout, variant, stages, swizzle_stride,
                              _stream()), "lc_hgemm_f16")
    return c


def gemm_fp8(a8, b8_nk, c, alpha=1.0, swizzle_stride=1):
    """c[M,N] (fp16) = alpha * a8[M,K] @ b8_nk[N,K]^T; a8/b8 are torch.float8_e4m3fn (or uint8 views)."""
    _need_gpu(a8, b8_nk, c)
    M, N, K = _gemm_dims(a8, b8_nk, c, b_is_nk=True)
    check(load().lc_gemm_fp8_e4m3(_ptr(a8), _ptr(b8_nk), _ptr(c), M, N, K, float(alpha), swizzle_stride, _stream()),
          "lc_gemm_fp8_e4m3")
    return c


def mxfp8_pack_scales(s):
    """s: uint8 [rows, K/32] E8M0 block scales -> the packed array lc_gemm_mxfp8 consumes (uint8 tensor of the same size, opaque)."""
    import torch
    _need_gpu(s)
    assert s.dtype == torch.uint8 and s.dim() == 2 and s.is_contiguous()
    rows, kb = s.shape
    p = torch.empty(rows * kb, dtype=torch.uint8, device=s.device)
    check(load().lc_mxfp8_pack_scales(_ptr(s), _ptr(p), rows, kb * 32, _stream()), "lc_mxfp8_pack_scales")
    return p


def gemm_mxfp8(a8, pa, b8_nk, pb, c, alpha=1.0, swizzle_stride=1):
    """c[M,N] (fp16) = alpha * (a8 * 2^(sa-127)) @ (b8_nk * 2^(sb-127))^T with per-(row, 32 k) E8M0 scales; pa / pb = mxfp8_pack_scales(sa / sb)."""
    _need_gpu(a8, b8_nk, c, pa, pb)
    M, N, K = _gemm_dims(a8, b8_nk, c, b_is_nk=True)
    assert pa.numel() == M * K // 32 and pb.numel() == N * K // 32
    check(load().lc_gemm_mxfp8(_ptr(a8), _ptr(pa), _ptr(b8_nk), _ptr(pb), _ptr(c), M, N, K, float(alpha), swizzle_stride, _stream()),
          "lc_gemm_mxfp8")
    return c


def hgemm_call(entry: str, a, b, c, stages=2, swizzle=False, swizzle_stride=1):
    _need_gpu(a, b, c)
    lay = C.c_int(LAYOUT_NN)
    load().lc_hgemm_entry_info(entry.encode(), C.byref(lay), None)   # unknown names fail in lc_hgemm_call below
    M, N, K = _gemm_dims(a, b, c, lay.value)
    check(load().lc_hgemm_call(entry.encode(), _ptr(a), _ptr(b), _ptr(c), M, N, K, stages, int(swizzle),
                               swizzle_stride, _stream()), entry)
    return c


def hgemm_vendor(a, b, c, layout=LAYOUT_NN):
    _need_gpu(a, b, c)
    M, N, K = _gemm_dims(a, b, c, layout)
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
    M, N, K = _gemm_dims(a, b, c, layout)
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
    B, H, N, D = _attn_dims(q, k, v, o, v_transposed)
    check(load().lc_attn_fwd_f16(_ptr(q), _ptr(k), _ptr(v), _ptr(o), B, H, N, D, int(v_transposed), family,
                                 int(acc_f32), stages, _stream()), "lc_attn_fwd_f16")
    return o


def attn_fwd_bf16(q, k, v, o):
    """bfloat16 forward for D in {256, 512} (BASELINE config 5 extension)."""
    import torch
    _need_gpu(q, k, v, o)
    assert q.dtype == k.dtype == v.dtype == o.dtype == torch.bfloat16
    B, H, N, D = _attn_dims(q, k, v, o)
    check(load().lc_attn_fwd_bf16(_ptr(q), _ptr(k), _ptr(v), _ptr(o), B, H, N, D, _stream()), "lc_attn_fwd_bf16")
    return o


def attn_call(entry: str, q, k, v, o, stages=2):
    _need_gpu(q, k, v, o)
    vt = C.c_int(0)
    load().lc_attn_entry_info(entry.encode(), None, C.byref(vt), None, None, None, None)
    B, H, N, D = _attn_dims(q, k, v, o, bool(vt.value))
    check(load().lc_attn_call(entry.encode(), _ptr(q), _ptr(k), _ptr(v), _ptr(o), B, H, N, D, stages,
                              _stream()), entry)
    return o


def attn_time(q, k, v, o, v_transposed=False, family=ATTN_SPLIT_Q, stages=2, warmup=1, iters=5) -> float:
    _need_gpu(q, k, v, o)
    B, H, N, D = _attn_dims(q, k, v, o, v_transposed)
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


# ---- measurement ------------------------------------------------------------------------------------
class Timer:
    """HIP events on the launch stream around any sequence of ABI launches (lc_timer_start / lc_timer_stop)."""

    def __enter__(self):
        self._t = C.c_void_p()
        check(load().lc_timer_start(_stream(), C.byref(self._t)), "lc_timer_start")
        self.ms = None
        return self

    def __exit__(self, et, ev, tb):
        ms = C.c_float(0)
        rc = load().lc_timer_stop(self._t, C.byref(ms))
        if et is None:
            check(rc, "lc_timer_stop")
            self.ms = ms.value
        return False


def hgemm_kernel_name(M, N, K, layout=LAYOUT_NN, variant=HGEMM_AUTO) -> str:
    buf = C.create_string_buffer(128)
    check(load().lc_hgemm_kernel_name(M, N, K, layout, variant, buf, 128), "lc_hgemm_kernel_name")
    return buf.value.decode()


def attn_slowpath_stats(reset=True):
    """[executions, sum of half-tile indices, non-finite, last offending row sum (float)] of attn_w4n's overflow slow path."""
    import struct
    out = (C.c_uint * 4)()
    check(load().lc_attn_slowpath_stats(out, int(reset)), "lc_attn_slowpath_stats")
    return [out[0], out[1], out[2], struct.unpack("f", struct.pack("I", out[3]))[0]]


def attn_kernel_name(N, D, v_transposed=False, bf16=False, bh=None) -> str:
    """The kernel the dispatcher picks for sequence length N and head dim D; bh = batch x heads of the launch (None: a grid that fills
    the GPU — the choice depends on it for D = 256 only)."""
    buf = C.create_string_buffer(128)
    check(load().lc_attn_kernel_name_bh(int(bh) if bh else -1, N, D, int(v_transposed), int(bf16), buf, 128), "lc_attn_kernel_name_bh")
    return buf.value.decode()


def clock_probe(out_u64x2):
    """Enqueue the {shader cycles, 100 MHz ticks} probe on the current stream (out: int64 cuda tensor of 2)."""
    check(load().lc_clock_probe(_ptr(out_u64x2), _stream()), "lc_clock_probe")
