// tu_attn_big4.hip — translation unit of the D = 1024 pair kernel (attn_bigd4.hip) and of attn_bigd2's V-transposed instantiation
// (D = 256, V as [B,H,D,N]: the reference's *_swizzle_qkv entries reach d = 256) — see lc_launch.h
#include <math.h>

#include "lc_launch.h"
#include "attn_bigd4.hip"

namespace lc {
// D = 1024, N % 64 == 0, V as [B,H,N,D], fp16
namespace {
template <int SP8>
int launch_bigd4_t(const half_t* Q, const half_t* K, const half_t* V, half_t* O, int B, int H, int N, hipStream_t st) {
  auto kern = attn_fwd_bigd4_kernel<SP8>;
  if (int rc = set_dyn_lds(kern, BD4_LDS)) return rc;
  const int nqb = N / 64;
  const dim3 grid((unsigned)((size_t)nqb * B * H)), block(256);
  const float sl2 = (1.0f / sqrtf(1024.0f)) * 1.4426950408889634f;
  hipLaunchKernelGGL(kern, grid, block, BD4_LDS, st, Q, K, V, O, N, (g_tune_attn_bigd_map == 1 ? nqb : -nqb)   /* auto = round-robin over the XCDs: + 3.7 % at twice the fabric bytes (MALL-resident K / V, L2 requests spread), profiles/r5f_bigd_map.log */, sl2,
                     /* KV-walk stagger by XCD: auto = with the round-robin map (+ 1.8 ... 2.1 %, profiles/r5h_bigd_stagger.log; nothing with the contiguous one) */
                     (g_tune_attn_bigd_stagger == 2 || (g_tune_attn_bigd_stagger == 0 && g_tune_attn_bigd_map != 1)) ? 1 : 0);
  return check_launch();
}
}  // namespace
int launch_attn_bigd4(const half_t* Q, const half_t* K, const half_t* V, half_t* O, int B, int H, int N, int span8, hipStream_t st) {
  // default: a batch's 8 pieces spread over its whole half-phase — with one and a half phases between issue and need
  // (attn_bigd4.hip: half-tile recycling) nothing is gained by issuing early, and the texture-address FIFO likes the pieces apart
  // (profiles/r4i_bigd4_v2.log: 8/8 867, 6/8 862, 4/8 845, 2/8 842 TFLOP/s; the first version of the kernel, with one phase per
  // piece, preferred 2/8: 733 vs 712)
  if (span8 == 2) return launch_bigd4_t<2>(Q, K, V, O, B, H, N, st);
  if (span8 == 4) return launch_bigd4_t<4>(Q, K, V, O, B, H, N, st);
  if (span8 == 6) return launch_bigd4_t<6>(Q, K, V, O, B, H, N, st);
  return launch_bigd4_t<8>(Q, K, V, O, B, H, N, st);
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
