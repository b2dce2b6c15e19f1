// tu_attn_big.hip — translation unit of the full-width large-head-dim attention kernel (attn_bigd2.hip) — see lc_launch.h
#include <math.h>

#include "lc_launch.h"
#include "attn_bigd2.hip"
#include "attn_bigd3.hip"

namespace lc {
namespace {
template <int D, bool BF16>
int launch_bigd2_t(const half_t* Q, const half_t* K, const half_t* V, half_t* O, int B, int H, int N, hipStream_t st) {
  auto kern = attn_fwd_bigd2_kernel<D, BF16>;
  constexpr int lds = bigd2_lds_bytes<D>();
  if (int rc = set_dyn_lds(kern, lds)) return rc;
  const int nqb = N / 128;
  const dim3 grid((unsigned)((size_t)nqb * B * H)), block(256);
  const float sl2 = (1.0f / sqrtf((float)D)) * 1.4426950408889634f;
  hipLaunchKernelGGL(kern, grid, block, lds, st, Q, K, V, O, N, nqb, sl2);
  return check_launch();
}
template <int D, bool BF16>
int launch_bigd3_t(const half_t* Q, const half_t* K, const half_t* V, half_t* O, int B, int H, int N, hipStream_t st) {
  auto kern = attn_fwd_bigd3_kernel<D, BF16>;
  constexpr int lds = bigd3_lds_bytes<D>();
  if (int rc = set_dyn_lds(kern, lds)) return rc;
  const int nqb = N / 128;
  const dim3 grid((unsigned)((size_t)nqb * B * H)), block(256);
  const float sl2 = (1.0f / sqrtf((float)D)) * 1.4426950408889634f;
  hipLaunchKernelGGL(kern, grid, block, lds, st, Q, K, V, O, N, nqb, sl2);
  return check_launch();
}
}  // namespace

// D in {256, 512}, N % 128 == 0, V as [B,H,N,D]
int launch_attn_bigd2(const half_t* Q, const half_t* K, const half_t* V, half_t* O, int B, int H, int N, int D, bool bf16,
                      hipStream_t st) {
  if (g_tune_attn_d512 == 2) {   // experimental: 32-row double-buffered tiles (attn_bigd3.hip)
    if (D == 512) return bf16 ? launch_bigd3_t<512, true>(Q, K, V, O, B, H, N, st) : launch_bigd3_t<512, false>(Q, K, V, O, B, H, N, st);
    if (D == 256) return bf16 ? launch_bigd3_t<256, true>(Q, K, V, O, B, H, N, st) : launch_bigd3_t<256, false>(Q, K, V, O, B, H, N, st);
  }
  if (D == 512) return bf16 ? launch_bigd2_t<512, true>(Q, K, V, O, B, H, N, st) : launch_bigd2_t<512, false>(Q, K, V, O, B, H, N, st);
  if (D == 256) return bf16 ? launch_bigd2_t<256, true>(Q, K, V, O, B, H, N, st) : launch_bigd2_t<256, false>(Q, K, V, O, B, H, N, st);
  return LC_ERR_HEADDIM;
}
}  // namespace lc
