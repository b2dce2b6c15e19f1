// mid256_probe.hip (liblc_diag.so only) — hgemm_mid_kernel's rotated hand-ordered loop on a 256 x 256 tile with two ring slots: the question
// "is that loop as good as hgemm_w4y's generated one?" priced before building a deeper (k-step-slot) ring on it.
// Not a product kernel: its accumulators fill the AGPR file, hipcc spills around the loops (scratch in prologue / epilogue only).
#include "../lc_launch.h"
#include "../hgemm_mid.hip"

extern "C" int lc_probe_mid256(const void* A, const void* B, void* C, int M, int N, int K, int b_kn, int panel_w, void* stream) {
  using namespace lc;
  if (M % 256 || N % 256 || K % 64 || K < 64) return -2;
  using G = Mid<4, 4, 2>;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int tiles_m = M / 256, tiles_n = N / 256;
  if (b_kn) {
    auto kern = hgemm_mid_kernel<true, 4, 4, 2>;
    if (int rc = set_dyn_lds(kern, G::LDS)) return rc;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), G::LDS, st, (const half_t*)A, (const half_t*)B, (half_t*)C, M, N, K, tiles_m, tiles_n, panel_w, -1);
  } else {
    auto kern = hgemm_mid_kernel<false, 4, 4, 2>;
    if (int rc = set_dyn_lds(kern, G::LDS)) return rc;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), G::LDS, st, (const half_t*)A, (const half_t*)B, (half_t*)C, M, N, K, tiles_m, tiles_n, panel_w, -1);
  }
  return hipGetLastError() == hipSuccess ? 0 : -4;
}
