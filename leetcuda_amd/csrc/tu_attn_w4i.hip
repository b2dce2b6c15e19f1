// tu_attn_w4i.hip — translation unit of the generated merged-phase attention kernel (attn_w4i.hip: one generated hand-ordered asm
// statement per phase) — see lc_launch.h
#include <math.h>

#include "lc_launch.h"
#define LC_AN_SLOWPATH_SYM g_ag_slowpath
#include "attn_w4i.hip"

namespace lc {
template <int D, int SCHED>
int launch_w4i_t(const half_t* Q, const half_t* K, const half_t* V, half_t* O, int B, int H, int N, hipStream_t st) {
  const int nqb = N / 256;
  const dim3 grid((unsigned)((size_t)nqb * B * H)), block(256);
  const float sl2 = (1.0f / sqrtf((float)D)) * 1.4426950408889634f;
  auto kern = attn_fwd_w4i_kernel<D, SCHED>;
  if (int rc = set_dyn_lds(kern, W4G<D>::LDS)) return rc;
  hipLaunchKernelGGL(kern, grid, block, W4G<D>::LDS, st, Q, K, V, O, N, nqb, sl2);
  return check_launch();
}
int launch_attn_w4i(const half_t* Q, const half_t* K, const half_t* V, half_t* O, int B, int H, int N, int D, int sched, hipStream_t st) {
  if (D == 32) return sched ? launch_w4i_t<32, 1>(Q, K, V, O, B, H, N, st) : launch_w4i_t<32, 0>(Q, K, V, O, B, H, N, st);
  if (D == 64) return sched ? launch_w4i_t<64, 1>(Q, K, V, O, B, H, N, st) : launch_w4i_t<64, 0>(Q, K, V, O, B, H, N, st);
  if (D == 96) return sched ? launch_w4i_t<96, 1>(Q, K, V, O, B, H, N, st) : launch_w4i_t<96, 0>(Q, K, V, O, B, H, N, st);
  if (D == 128) return sched ? launch_w4i_t<128, 1>(Q, K, V, O, B, H, N, st) : launch_w4i_t<128, 0>(Q, K, V, O, B, H, N, st);
  return LC_ERR_HEADDIM;
}

// slow-path counters of THIS unit's kernels, added onto out4[0..2] (out4[3]: last offender, taken when this unit has one)
int diag_attn_slowpath_g(unsigned* out4, int reset) {
  unsigned mine[4] = {0, 0, 0, 0};
  if (hipMemcpyFromSymbol(mine, HIP_SYMBOL(g_ag_slowpath), 16) != hipSuccess) return LC_ERR_LAUNCH;
  if (out4) {
    for (int i = 0; i < 3; ++i) out4[i] += mine[i];
    if (mine[0]) out4[3] = mine[3];
  }
  if (reset) {
    const unsigned z[4] = {0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_ag_slowpath), z, 16) != hipSuccess) return LC_ERR_LAUNCH;
  }
  return LC_OK;
}
}  // namespace lc
