// tu_attn_w4p.hip — translation unit of the persistent merged-phase attention kernel (attn_w4p.hip) — see lc_launch.h
#include <math.h>

#include "lc_launch.h"
#define LC_AN_SLOWPATH_SYM g_ap_slowpath
#include "attn_w4p.hip"

namespace lc {
namespace {
template <int D>
int launch_w4p_t(const half_t* Q, const half_t* K, const half_t* V, half_t* O, int B, int H, int N, hipStream_t st) {
  const int nqb = N / 256;
  const size_t nblk = (size_t)nqb * B * H;
  const int ncu = device_cu_count();   // one workgroup per CU: each takes a CU's whole register file and > half its LDS
  const dim3 grid((unsigned)(nblk < (size_t)ncu ? nblk : (size_t)ncu)), block(256);
  const float sl2 = (1.0f / sqrtf((float)D)) * 1.4426950408889634f;
  auto kern = attn_fwd_w4p_kernel<D>;
  if (int rc = set_dyn_lds(kern, W4P<D>::LDS)) return rc;
  hipLaunchKernelGGL(kern, grid, block, W4P<D>::LDS, st, Q, K, V, O, N, nqb, sl2, (int)nblk, (int)grid.x);
  return check_launch();
}
}  // namespace

// D in {64, 128}, N % 256 == 0, V as [B,H,N,D]
int launch_attn_w4p(const half_t* Q, const half_t* K, const half_t* V, half_t* O, int B, int H, int N, int D, hipStream_t st) {
  if (D == 64) return launch_w4p_t<64>(Q, K, V, O, B, H, N, st);
  if (D == 128) return launch_w4p_t<128>(Q, K, V, O, B, H, N, st);
  return LC_ERR_HEADDIM;
}
// slow-path counters of THIS unit's kernels, added onto out4[0..2] (out4[3]: last offender, taken when this unit has one)
int diag_attn_slowpath_p(unsigned* out4, int reset) {
  unsigned mine[4] = {0, 0, 0, 0};
  if (hipMemcpyFromSymbol(mine, HIP_SYMBOL(g_ap_slowpath), 16) != hipSuccess) return LC_ERR_LAUNCH;
  if (out4) {
    for (int i = 0; i < 3; ++i) out4[i] += mine[i];
    if (mine[0]) out4[3] = mine[3];
  }
  if (reset) {
    const unsigned z[4] = {0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_ap_slowpath), z, 16) != hipSuccess) return LC_ERR_LAUNCH;
  }
  return LC_OK;
}
}  // namespace lc
