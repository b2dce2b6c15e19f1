// tu_fp8k.hip — translation unit of the K = 128 MX fp8 GEMM kernel (gemm_fp8_w4k.hip) — see lc_launch.h
#include "lc_launch.h"
#include "gemm_fp8_w4k.hip"

namespace lc {
namespace {
template <bool MX>
int launch_w4k(const uint8_t* A, const uint8_t* B, half_t* C, int M, int N, int K, float alpha, int tiles_m, int tiles_n, int pw,
               const uint32_t* PA, const uint32_t* PB, hipStream_t st) {
  auto kern = gemm_fp8_w4k_kernel<MX>;
  if (int rc = set_dyn_lds(kern, W4B_LDS)) return rc;
  // persistent workgroups under hgemm_w4y's rule (lc_tune_set "hgemm_persist"): one per CU when every one gets the same number of tiles
  const int nblk = tiles_m * tiles_n, ncu = device_cu_count();
  const bool persist = g_tune_hgemm_persist != 0 && nblk > ncu && nblk % ncu == 0;
  hipLaunchKernelGGL(kern, dim3(persist ? ncu : nblk), dim3(256), W4B_LDS, st, A, B, C, M, N, K, alpha, tiles_m, tiles_n, pw,
                     stagger_arg(K / BK8K), persist ? nblk : 0, PA, PB);
  return check_launch();
}
}  // namespace

bool gemm_fp8_w4k_fits(int K) { return (size_t)K * 260 < ((size_t)1 << 31); }   // 32-bit DMA offsets: a wave's pieces reach 232 rows past its base

int launch_gemm_fp8_w4k(const uint8_t* A, const uint8_t* B, half_t* C, int M, int N, int K, float alpha, int tiles_m, int tiles_n, int pw,
                        hipStream_t st) {
  return launch_w4k<false>(A, B, C, M, N, K, alpha, tiles_m, tiles_n, pw, nullptr, nullptr, st);
}

int launch_gemm_mxfp8(const uint8_t* A, const uint32_t* PA, const uint8_t* B, const uint32_t* PB, half_t* C, int M, int N, int K, float alpha,
                      int tiles_m, int tiles_n, int pw, hipStream_t st) {
  return launch_w4k<true>(A, B, C, M, N, K, alpha, tiles_m, tiles_n, pw, PA, PB, st);
}

int launch_mx_pack_scales(const uint8_t* S, uint32_t* P, int rows, int K, hipStream_t st) {
  const size_t total = (size_t)(rows / 128) * (K / BK8K) * 128;
  hipLaunchKernelGGL(mx_pack_scales_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, S, P, rows, K);
  return check_launch();
}
}  // namespace lc
