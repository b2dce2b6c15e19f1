// tu_w4.hip — translation unit of the 4-wave HGEMM kernels (hgemm_w4.hip) — see lc_launch.h
#include "lc_launch.h"
#include "hgemm_w4.hip"

namespace lc {
namespace {
template <bool B_KN>
int launch_w4_t(const half_t* A, const half_t* B, half_t* C, int M, int N, int K, int variant, int tiles_m,
                int tiles_n, int pw, hipStream_t st) {
  const dim3 grid(tiles_m * tiles_n);
  // buffer-descriptor DMA addresses are 32-bit offsets from the wave's first row: fall back to the 64-bit global form
  // when an offset could reach 2 GiB (NN: K tiles step through the whole of B)
  if (variant == LC_HGEMM_MFMA256W4C) {
    const size_t max_off = B_KN ? (size_t)K * N * 2 + (size_t)N * 64 : (size_t)K * 2 * 130;
    if (max_off >= ((size_t)1 << 31)) variant = LC_HGEMM_MFMA256W4B;
  }
#ifdef LC_DIAG
  if (variant == LC_HGEMM_MFMA256W4C && g_tune_hgemm_stamps) {
    auto kern = hgemm_w4b_kernel<B_KN, true, true>;
    if (int rc = set_dyn_lds(kern, W4B_LDS)) return rc;
    hipLaunchKernelGGL(kern, grid, dim3(256), W4B_LDS, st, A, B, C, M, N, K, tiles_m, tiles_n, pw);
  } else
#endif
  if (variant == LC_HGEMM_MFMA256W4C) {
    auto kern = hgemm_w4b_kernel<B_KN, true>;
    if (int rc = set_dyn_lds(kern, W4B_LDS)) return rc;
    hipLaunchKernelGGL(kern, grid, dim3(256), W4B_LDS, st, A, B, C, M, N, K, tiles_m, tiles_n, pw);
  } else if (variant == LC_HGEMM_MFMA256W4B) {
    auto kern = hgemm_w4b_kernel<B_KN>;
    if (int rc = set_dyn_lds(kern, W4B_LDS)) return rc;
    hipLaunchKernelGGL(kern, grid, dim3(256), W4B_LDS, st, A, B, C, M, N, K, tiles_m, tiles_n, pw);
  } else if (variant == LC_HGEMM_MFMA256W4S) {
    auto kern = hgemm_w4s_kernel<B_KN>;
    if (int rc = set_dyn_lds(kern, HGEMM256_LDS)) return rc;
    hipLaunchKernelGGL(kern, grid, dim3(256), HGEMM256_LDS, st, A, B, C, M, N, K, tiles_m, tiles_n, pw);
  } else {
#define LC_W4_CASE(ABL)                                                                                   \
  case ABL: {                                                                                             \
    auto kern = hgemm_w4_kernel<B_KN, ABL>;                                                               \
    if (int rc = set_dyn_lds(kern, HGEMM256_LDS)) return rc;                                              \
    hipLaunchKernelGGL(kern, grid, dim3(256), HGEMM256_LDS, st, A, B, C, M, N, K, tiles_m, tiles_n, pw);  \
  } break;
    switch (g_tune_w4_abl) {
      LC_W4_CASE(0)
#ifdef LC_DIAG
      LC_W4_CASE(1) LC_W4_CASE(2) LC_W4_CASE(3) LC_W4_CASE(4) LC_W4_CASE(7)
#endif
      default: return LC_ERR_ARG;
    }
#undef LC_W4_CASE
  }
  return check_launch();
}
}  // namespace

int launch_w4_family(const half_t* A, const half_t* B, half_t* C, int M, int N, int K, int variant, bool b_kn,
                     int tiles_m, int tiles_n, int panel_w, hipStream_t st) {
  return b_kn ? launch_w4_t<true>(A, B, C, M, N, K, variant, tiles_m, tiles_n, panel_w, st)
              : launch_w4_t<false>(A, B, C, M, N, K, variant, tiles_m, tiles_n, panel_w, st);
}
}  // namespace lc
