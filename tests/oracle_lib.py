"""ctypes loader for oracle/liblc_oracle.so — the CHECKER. Imported only by tests/, smoke() and the
cpu_baseline leg of bench.py (see oracle/lc_oracle.h)."""
import ctypes as C
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
PATH = ROOT / "oracle" / "liblc_oracle.so"
_u16 = np.ctypeslib.ndpointer(dtype=np.uint16, flags="C_CONTIGUOUS")
_f32 = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_i = C.c_int


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        lib.lc_h2f.restype, lib.lc_h2f.argtypes = C.c_float, [C.c_uint16]
        lib.lc_f2h.restype, lib.lc_f2h.argtypes = C.c_uint16, [C.c_float]
        lib.lc_d2h.restype, lib.lc_d2h.argtypes = C.c_uint16, [C.c_double]
        for n, out in (("lc_oracle_hgemm_exact", _u16), ("lc_oracle_hgemm_exact_f32", _f32),
                       ("lc_oracle_hgemm_refnum", _u16)):
            f = getattr(lib, n)
            f.restype, f.argtypes = None, [_u16, _u16, out, _i, _i, _i, _i]
        for n, out in (("lc_oracle_attn_exact", _u16), ("lc_oracle_attn_exact_f32", _f32)):
            f = getattr(lib, n)
            f.restype, f.argtypes = None, [_u16, _u16, _u16, out, _i, _i, _i, _i, _i]
        lib.lc_oracle_attn_exact_f32_rows.restype = None
        lib.lc_oracle_attn_exact_f32_rows.argtypes = [_u16, _u16, _u16, _f32, _i, _i, _i, _i, _i]
        lib.lc_oracle_attn_refnum.restype = None
        lib.lc_oracle_attn_refnum.argtypes = [_u16, _u16, _u16, _u16, _i, _i, _i, _i, _i, _i]
        lib.lc_oracle_hgemm_flops.restype, lib.lc_oracle_hgemm_flops.argtypes = C.c_double, [_i, _i, _i]
        lib.lc_oracle_mha_flops.restype, lib.lc_oracle_mha_flops.argtypes = C.c_double, [_i] * 5
        lib.lc_oracle_block_swizzle_stride.restype = _i
        lib.lc_oracle_block_swizzle_stride.argtypes = [_i, _i, C.c_double]
        lib.lc_oracle_num_threads.restype, lib.lc_oracle_num_threads.argtypes = _i, []
        lib.lc_oracle_attn_exact_f32_bf16.restype = None
        lib.lc_oracle_attn_exact_f32_bf16.argtypes = [_u16, _u16, _u16, _f32, _i, _i, _i, _i]
        lib.lc_oracle_attn_exact_f32_rows_bf16.restype = None
        lib.lc_oracle_attn_exact_f32_rows_bf16.argtypes = [_u16, _u16, _u16, _f32, _i, _i, _i, _i]
        lib.lc_e4m3_to_f32.restype, lib.lc_e4m3_to_f32.argtypes = C.c_float, [C.c_uint8]
        _u8 = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
        lib.lc_oracle_gemm_fp8_exact_f32.restype = None
        lib.lc_oracle_gemm_fp8_exact_f32.argtypes = [_u8, _u8, _f32, _i, _i, _i, C.c_float]
        lib.lc_oracle_gemm_mxfp8_exact_f32.restype = None
        lib.lc_oracle_gemm_mxfp8_exact_f32.argtypes = [_u8, _u8, _u8, _u8, _f32, _i, _i, _i, C.c_float]

    # ---- numpy-level helpers (uint16 views of fp16 data) -------------------------------------
    @staticmethod
    def u16(x):
        """torch fp16 tensor / numpy fp16 array -> contiguous uint16 numpy array."""
        if hasattr(x, "detach"):
            x = x.detach().cpu().contiguous().numpy()
        x = np.ascontiguousarray(x)
        return x.view(np.uint16) if x.dtype == np.float16 else x.astype(np.uint16, copy=False)

    def hgemm(self, a, b, M, N, K, layout=0, mode="exact"):
        a, b = self.u16(a), self.u16(b)
        if mode == "f32":
            c = np.empty((M, N), np.float32)
            self.lib.lc_oracle_hgemm_exact_f32(a, b, c, M, N, K, layout)
            return c
        c = np.empty((M, N), np.uint16)
        fn = self.lib.lc_oracle_hgemm_exact if mode == "exact" else self.lib.lc_oracle_hgemm_refnum
        fn(a, b, c, M, N, K, layout)
        return c.view(np.float16)

    def gemm_fp8(self, a8, b8_nk, M, N, K, alpha=1.0):
        """a8 [M,K], b8_nk [N,K]: uint8 views of e4m3fn data -> fp32 exact result."""
        if hasattr(a8, "detach"):
            a8 = a8.detach().cpu().contiguous().view(__import__("torch").uint8).numpy()
            b8_nk = b8_nk.detach().cpu().contiguous().view(__import__("torch").uint8).numpy()
        c = np.empty((M, N), np.float32)
        self.lib.lc_oracle_gemm_fp8_exact_f32(np.ascontiguousarray(a8), np.ascontiguousarray(b8_nk), c, M, N, K,
                                              float(alpha))
        return c

    def gemm_mxfp8(self, a8, sa, b8_nk, sb, M, N, K, alpha=1.0):
        """a8 [M,K], b8_nk [N,K] e4m3fn (uint8 views); sa [M,K/32], sb [N,K/32] E8M0 -> fp32 exact result."""
        def u8(x):
            if hasattr(x, "detach"):
                x = x.detach().cpu().contiguous().view(__import__("torch").uint8).numpy()
            return np.ascontiguousarray(x, dtype=np.uint8)
        c = np.empty((M, N), np.float32)
        self.lib.lc_oracle_gemm_mxfp8_exact_f32(u8(a8), u8(sa), u8(b8_nk), u8(sb), c, M, N, K, float(alpha))
        return c

    def attn(self, q, k, v, B, H, N, D, vt=False, mode="exact", Bc=64, o_f32=False):
        q, k, v = self.u16(q), self.u16(k), self.u16(v)
        if mode == "f32":
            o = np.empty((B, H, N, D), np.float32)
            self.lib.lc_oracle_attn_exact_f32(q, k, v, o, B, H, N, D, int(vt))
            return o
        o = np.empty((B, H, N, D), np.uint16)
        if mode == "exact":
            self.lib.lc_oracle_attn_exact(q, k, v, o, B, H, N, D, int(vt))
        else:
            self.lib.lc_oracle_attn_refnum(q, k, v, o, B, H, N, D, Bc, int(o_f32))
        return o.view(np.float16)

    def attn_bf16(self, q, k, v, B, H, N, D):
        """torch.bfloat16 tensors -> fp32 exact attention."""
        import torch
        def u(x):
            return np.ascontiguousarray(x.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16))
        o = np.empty((B, H, N, D), np.float32)
        self.lib.lc_oracle_attn_exact_f32_bf16(u(q), u(k), u(v), o, B, H, N, D)
        return o

    def attn_rows_bf16(self, qrows, k, v, BH, Nq, N, D):
        """torch.bfloat16 tensors: Nq sampled query rows per problem against all N keys -> fp32 exact."""
        import torch
        def u(x):
            return np.ascontiguousarray(x.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16))
        o = np.empty((BH, Nq, D), np.float32)
        self.lib.lc_oracle_attn_exact_f32_rows_bf16(u(qrows), u(k), u(v), o, BH, Nq, N, D)
        return o

    def attn_rows(self, qrows, k, v, BH, Nq, N, D, vt=False):
        o = np.empty((BH, Nq, D), np.float32)
        self.lib.lc_oracle_attn_exact_f32_rows(self.u16(qrows), self.u16(k), self.u16(v), o, BH, Nq, N, D,
                                               int(vt))
        return o


def load() -> Oracle:
    if not PATH.exists():
        raise RuntimeError(f"{PATH} missing: run `python -m leetcuda_amd.build`")
    return Oracle(C.CDLL(str(PATH)))
