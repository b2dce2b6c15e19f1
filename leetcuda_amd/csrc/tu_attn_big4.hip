// tu_attn_big4.hip — translation unit of the D = 1024 pair kernel (attn_bigd4.hip) and of attn_bigd2's V-transposed instantiation
// (D = 256, V as [B,H,D,N]: the reference's *_swizzle_qkv entries reach d = 256) — see lc_launch.h
#include <math.h>

#include "lc_launch.h"
#include "attn_bigd4.hip"

namespace lc {
// D = 1024, N % 64 == 0, V as [B,H,N,D], fp16
int launch_attn_bigd4(const half_t* Q, const half_t* K, const half_t* V, half_t* O, int B, int H, int N, hipStream_t st) {
  auto kern = attn_fwd_bigd4_kernel;
  if (int rc = set_dyn_lds(kern, BD4_LDS)) return rc;
  const int nqb = N / 64;
  const dim3 grid((unsigned)((size_t)nqb * B * H)), block(256);
  const float sl2 = (1.0f / sqrtf(1024.0f)) * 1.4426950408889634f;
  hipLaunchKernelGGL(kern, grid, block, BD4_LDS, st, Q, K, V, O, N, nqb, sl2);
  return check_launch();
}
// D = 256, N % 128 == 0, V as [B,H,D,N], fp16
int launch_attn_bigd2_vt(const half_t* Q, const half_t* K, const half_t* V, half_t* O, int B, int H, int N, int D, hipStream_t st) {
  if (D != 256) return LC_ERR_HEADDIM;
  auto kern = attn_fwd_bigd2_kernel<256, false, true>;
  constexpr int lds = bigd2_lds_bytes<256>();
  if (int rc = set_dyn_lds(kern, lds)) return rc;
  const int nqb = N / 128;
  const dim3 grid((unsigned)((size_t)nqb * B * H)), block(256);
  const float sl2 = (1.0f / sqrtf(256.0f)) * 1.4426950408889634f;
  hipLaunchKernelGGL(kern, grid, block, lds, st, Q, K, V, O, N, nqb, sl2);
  return check_launch();
}
}  // namespace lc
