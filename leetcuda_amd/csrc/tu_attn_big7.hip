// tu_attn_big7.hip — translation unit of attn_bigd7.hip (D = 256, 64 query rows per wave on v_mfma_f32_16x16x32, fp16 / bf16) — see lc_launch.h
#include <math.h>

#include "lc_launch.h"
#include "attn_bigd7.hip"

namespace lc {
namespace {
template <bool BF16, bool VT>
int launch_bigd7_t(const half_t* Q, const half_t* K, const half_t* V, half_t* O, int B, int H, int N, hipStream_t st) {
  auto kern = attn_fwd_bigd7_kernel<BF16, VT>;
  constexpr int lds = bd7_lds_bytes();
  if (int rc = set_dyn_lds(kern, lds)) return rc;
  const int nqb = (N + 255) / 256;   // (N % 256 == 128: the head's last block is half real)
  const dim3 grid((unsigned)((size_t)nqb * B * H)), block(256);
  const float sl2 = (1.0f / sqrtf(256.0f)) * 1.4426950408889634f;
  hipLaunchKernelGGL(kern, grid, block, lds, st, Q, K, V, O, N, nqb, sl2);
  return check_launch();
}
}  // namespace
// D = 256, N % 256 == 0 (or N % 256 == 128: last block half real), V as [B,H,N,D]; fp16 or bf16
int launch_attn_bigd7(const half_t* Q, const half_t* K, const half_t* V, half_t* O, int B, int H, int N, bool bf16, hipStream_t st) {
  return bf16 ? launch_bigd7_t<true, false>(Q, K, V, O, B, H, N, st) : launch_bigd7_t<false, false>(Q, K, V, O, B, H, N, st);
}
// the same with V as [B,H,D,N] (fp16: the reference's *_swizzle_qkv entries)
int launch_attn_bigd7_vt(const half_t* Q, const half_t* K, const half_t* V, half_t* O, int B, int H, int N, hipStream_t st) {
  return launch_bigd7_t<false, true>(Q, K, V, O, B, H, N, st);
}
}  // namespace lc
