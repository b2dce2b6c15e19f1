// tu_valu.hip — translation unit of the vector-ALU HGEMM ladder (hgemm_valu.hip) — see lc_launch.h
#include "lc_launch.h"
#include "hgemm_valu.hip"

namespace lc {
namespace {
template <int TM, int BKK, int VEC, bool PACK, bool BCF, bool DBUF>
int launch_valu_tile(const half_t* A, const half_t* B, half_t* C, int M, int N, int K, hipStream_t st) {
  hipLaunchKernelGGL((hgemm_valu_tile_kernel<TM, BKK, VEC, PACK, BCF, DBUF>), dim3(N / 128, M / (16 * TM)), dim3(256), 0, st, A, B,
                     C, M, N, K);
  return check_launch();
}
}  // namespace

// tile requirements of a rung: {rows, columns, k} multiples (0 = none)
void valu_rung_tile(int rung, int* tm, int* tn, int* tk) {
  switch (rung) {
    case LC_HGEMM_VALU_NAIVE: *tm = 1, *tn = 1, *tk = 1; break;
    case LC_HGEMM_VALU_SLICED_K: *tm = 32, *tn = 32, *tk = 32; break;
    case LC_HGEMM_VALU_T8X8_K16: *tm = 128, *tn = 128, *tk = 16; break;
    case LC_HGEMM_VALU_T8X8_K32: *tm = 128, *tn = 128, *tk = 32; break;
    case LC_HGEMM_VALU_T16X8_K32: *tm = 256, *tn = 128, *tk = 32; break;
    default: *tm = 128, *tn = 128, *tk = 8; break;
  }
}

const char* valu_rung_kernel_name(int rung) {
  switch (rung) {
    case LC_HGEMM_VALU_NAIVE: return "hgemm_valu_naive_kernel";
    case LC_HGEMM_VALU_SLICED_K: return "hgemm_valu_sliced_k_kernel";
    case LC_HGEMM_VALU_T8X8_X4: return "hgemm_valu_tile_kernel<8,8,4,false,false,false>";
    case LC_HGEMM_VALU_T8X8_X4_PACK: return "hgemm_valu_tile_kernel<8,8,4,true,false,false>";
    case LC_HGEMM_VALU_T8X8_X4_BCF: return "hgemm_valu_tile_kernel<8,8,4,false,true,false>";
    case LC_HGEMM_VALU_T8X8_X4_PACK_BCF: return "hgemm_valu_tile_kernel<8,8,4,true,true,false>";
    case LC_HGEMM_VALU_T8X8_X8_PACK_BCF: return "hgemm_valu_tile_kernel<8,8,8,true,true,false>";
    case LC_HGEMM_VALU_T8X8_X8_PACK_BCF_DBUF: return "hgemm_valu_tile_kernel<8,8,8,true,true,true>";
    case LC_HGEMM_VALU_T8X8_K16: return "hgemm_valu_tile_kernel<8,16,8,true,true,true>";
    case LC_HGEMM_VALU_T8X8_K32: return "hgemm_valu_tile_kernel<8,32,8,true,true,true>";
    case LC_HGEMM_VALU_T16X8_K32: return "hgemm_valu_tile_kernel<16,32,8,true,true,true>";
    default: return nullptr;
  }
}

int launch_valu_rung(const half_t* A, const half_t* B, half_t* C, int M, int N, int K, int rung, hipStream_t st) {
  switch (rung) {
    case LC_HGEMM_VALU_NAIVE:
      hipLaunchKernelGGL(hgemm_valu_naive_kernel, dim3((N + 63) / 64, (M + 3) / 4), dim3(256), 0, st, A, B, C, M, N, K);
      return check_launch();
    case LC_HGEMM_VALU_SLICED_K:
      hipLaunchKernelGGL(hgemm_valu_sliced_k_kernel, dim3(N / 32, M / 32), dim3(1024), 0, st, A, B, C, M, N, K);
      return check_launch();
    case LC_HGEMM_VALU_T8X8_X4: return launch_valu_tile<8, 8, 4, false, false, false>(A, B, C, M, N, K, st);
    case LC_HGEMM_VALU_T8X8_X4_PACK: return launch_valu_tile<8, 8, 4, true, false, false>(A, B, C, M, N, K, st);
    case LC_HGEMM_VALU_T8X8_X4_BCF: return launch_valu_tile<8, 8, 4, false, true, false>(A, B, C, M, N, K, st);
    case LC_HGEMM_VALU_T8X8_X4_PACK_BCF: return launch_valu_tile<8, 8, 4, true, true, false>(A, B, C, M, N, K, st);
    case LC_HGEMM_VALU_T8X8_X8_PACK_BCF: return launch_valu_tile<8, 8, 8, true, true, false>(A, B, C, M, N, K, st);
    case LC_HGEMM_VALU_T8X8_X8_PACK_BCF_DBUF: return launch_valu_tile<8, 8, 8, true, true, true>(A, B, C, M, N, K, st);
    case LC_HGEMM_VALU_T8X8_K16: return launch_valu_tile<8, 16, 8, true, true, true>(A, B, C, M, N, K, st);
    case LC_HGEMM_VALU_T8X8_K32: return launch_valu_tile<8, 32, 8, true, true, true>(A, B, C, M, N, K, st);
    case LC_HGEMM_VALU_T16X8_K32: return launch_valu_tile<16, 32, 8, true, true, true>(A, B, C, M, N, K, st);
    default: return LC_ERR_ARG;
  }
}
}  // namespace lc
