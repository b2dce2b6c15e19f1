// attn_w4u_stamps.hip — liblc_diag.so: attn_fwd_w4u_kernel<128, false, WALK = 0 / 3> compiled with W4U_STAMPS (attn_w4u.hip): wave 0 of workgroup 0
// records s_memtime at the milestones of its block.  tools/attn_w4u_stamps.py prints where a block's fixed cost goes.
#include <math.h>

#define W4U_STAMPS 1
#define LC_AN_SLOWPATH_SYM g_diag_w4u_stamps_slowpath
#include "../lc_launch.h"
#include "../attn_w4u.hip"

// nsplit == 1: one 256-row block per workgroup over the whole head (WALK 0, O = the output); nsplit >= 2: WALK 3, O = partials
// [nsplit][B H][N][128], lse = [nsplit][B H][N] (no combine: the stamps are the point).  out16: stamps 0 .. 11 (shader cycles), [14] / [15] =
// s_memrealtime (100 MHz) at entry / exit.
extern "C" int lc_diag_attn_w4u_stamps(const void* Q, const void* K, const void* V, void* O, void* lse, int B, int H, int N, int nsplit,
                                       void* out_u64x16, void* stream) {
  using namespace lc;
  constexpr int D = 128;
  if (!Q || !K || !V || !O || !out_u64x16 || N % 256 != 0 || nsplit < 1 || (N / 64) % nsplit != 0 || (N / 64) / nsplit < 2) return LC_ERR_ARG;
  if (nsplit > 1 && !lse) return LC_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const float sl2 = (1.0f / sqrtf((float)D)) * 1.4426950408889634f;
  const int nblk = (N / 256) * B * H * nsplit, lds = W4U<D>::LDS + 256;
  if (nsplit == 1) {
    auto kern = attn_fwd_w4u_kernel<D, false, 0>;
    if (int rc = set_dyn_lds(kern, lds)) return rc;
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), lds, st, (const half_t*)Q, (const half_t*)K, (const half_t*)V, (half_t*)O, N, N / 256, sl2, nblk, nblk,
                       0, 1, (float*)nullptr);
  } else {
    auto kern = attn_fwd_w4u_kernel<D, false, 3>;
    if (int rc = set_dyn_lds(kern, lds)) return rc;
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), lds, st, (const half_t*)Q, (const half_t*)K, (const half_t*)V, (half_t*)O, N, N / 256, sl2, nblk, nblk,
                       0, nsplit, (float*)lse);
  }
  if (int rc = check_launch()) return rc;
  if (hipStreamSynchronize(st) != hipSuccess) return LC_ERR_LAUNCH;
  return hipMemcpyFromSymbol(out_u64x16, HIP_SYMBOL(g_w4u_stamps), 16 * sizeof(unsigned long long)) == hipSuccess ? LC_OK : LC_ERR_LAUNCH;
}
