// tu_fp8.hip — translation unit of the fp8 GEMM kernels (gemm_fp8.hip) — see lc_launch.h
#include "lc_launch.h"
#include "gemm_fp8.hip"

namespace lc {
int launch_gemm_fp8(const uint8_t* A, const uint8_t* B, half_t* C, int M, int N, int K, float alpha, int tiles_m,
                    int tiles_n, int pw, int mx, hipStream_t st) {
  // MX + 4-wave kernel (mx = 1) unless a buffer offset could reach 2 GiB; mx = 2: MX 8-wave kernel; 0: plain K = 16
  if (mx == 1 && (size_t)K * 260 < ((size_t)1 << 31)) {   // (a wave's pieces reach 232 rows past its base)
    auto kern = gemm_fp8_w4_kernel;
    if (int rc = set_dyn_lds(kern, W4B_LDS)) return rc;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), W4B_LDS, st, A, B, C, M, N, K, alpha, tiles_m,
                       tiles_n, pw, stagger_arg(K / BK8));   // K-loop stagger as hgemm_w4y (lc_launch.h)
  } else if (mx) {
    auto kern = gemm_fp8_pingpong2_kernel<true>;
    if (int rc = set_dyn_lds(kern, HGEMM256_LDS)) return rc;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(512), HGEMM256_LDS, st, A, B, C, M, N, K, alpha, tiles_m,
                       tiles_n, pw);
  } else {
    auto kern = gemm_fp8_pingpong2_kernel<false>;
    if (int rc = set_dyn_lds(kern, HGEMM256_LDS)) return rc;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(512), HGEMM256_LDS, st, A, B, C, M, N, K, alpha, tiles_m,
                       tiles_n, pw);
  }
  return check_launch();
}
}  // namespace lc
