// tu_attn_w4.hip — translation unit of the 4-wave x 64-row merged-phase attention kernel (attn_w4m.hip) — see lc_launch.h
#include <math.h>

#include "lc_launch.h"
#include "attn_w4n.hip"

namespace lc {
// merged-phase kernel (attn_w4m.hip); pad = wait states appended to the accumulating Q·Kᵀ MFMAs (0 or 4, A/B knob)
int launch_attn_w4m_d128(const half_t* Q, const half_t* K, const half_t* V, half_t* O, int B, int H, int N, int pad,
                         hipStream_t st) {
  constexpr int D = 128;
  const int nqb = N / 256;
  const dim3 grid((unsigned)((size_t)nqb * B * H)), block(256);
  const float sl2 = (1.0f / sqrtf((float)D)) * 1.4426950408889634f;
  if (pad) {
    auto kern = attn_fwd_w4m_kernel<D, 4>;
    if (int rc = set_dyn_lds(kern, AM_LDS)) return rc;
    hipLaunchKernelGGL(kern, grid, block, AM_LDS, st, Q, K, V, O, N, nqb, sl2);
  } else {
    auto kern = attn_fwd_w4m_kernel<D, 0>;
    if (int rc = set_dyn_lds(kern, AM_LDS)) return rc;
    hipLaunchKernelGGL(kern, grid, block, AM_LDS, st, Q, K, V, O, N, nqb, sl2);
  }
  return check_launch();
}
int launch_attn_w4n_d128(const half_t* Q, const half_t* K, const half_t* V, half_t* O, int B, int H, int N, hipStream_t st) {
  constexpr int D = 128;
  const int nqb = N / 256;
  const dim3 grid((unsigned)((size_t)nqb * B * H)), block(256);
  const float sl2 = (1.0f / sqrtf((float)D)) * 1.4426950408889634f;
  auto kern = attn_fwd_w4n_kernel<D>;
  if (int rc = set_dyn_lds(kern, AM_LDS)) return rc;
  hipLaunchKernelGGL(kern, grid, block, AM_LDS, st, Q, K, V, O, N, nqb, sl2);
  return check_launch();
}
int diag_attn_slowpath(unsigned* out4, int reset) {
  if (out4 && hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_an_slowpath), 16) != hipSuccess) return LC_ERR_LAUNCH;
  if (reset) {
    const unsigned z[4] = {0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_an_slowpath), z, 16) != hipSuccess) return LC_ERR_LAUNCH;
  }
  return LC_OK;
}
}  // namespace lc
