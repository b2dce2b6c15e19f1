// tu_attn_big6.hip — translation unit of attn_bigd6.hip (D = 512 on v_mfma_f32_16x16x32, fp16 / bf16) — see lc_launch.h
#include <math.h>

#include "lc_launch.h"
#include "attn_bigd6.hip"

namespace lc {
namespace {
template <bool BF16>
int launch_bigd6_t(const half_t* Q, const half_t* K, const half_t* V, half_t* O, int B, int H, int N, hipStream_t st) {
  auto kern = attn_fwd_bigd6_kernel<BF16>;
  constexpr int lds = bigd2_lds_bytes<512>();
  if (int rc = set_dyn_lds(kern, lds)) return rc;
  const int nqb = N / 128;
  const dim3 grid((unsigned)((size_t)nqb * B * H)), block(256);
  const float sl2 = (1.0f / sqrtf(512.0f)) * 1.4426950408889634f;
  hipLaunchKernelGGL(kern, grid, block, lds, st, Q, K, V, O, N, (g_tune_attn_bigd_map == 2 ? -nqb : nqb)   /* auto = XCD-contiguous: round-robin measured - 2 % here, profiles/r5f_bigd_map.log */, sl2);
  return check_launch();
}
}  // namespace
// D = 512, N % 128 == 0, V as [B,H,N,D]; fp16 or bf16
int launch_attn_bigd6(const half_t* Q, const half_t* K, const half_t* V, half_t* O, int B, int H, int N, bool bf16, hipStream_t st) {
  return bf16 ? launch_bigd6_t<true>(Q, K, V, O, B, H, N, st) : launch_bigd6_t<false>(Q, K, V, O, B, H, N, st);
}
}  // namespace lc
