// attn_w4i_abl.hip — liblc_diag.so: ONE ablated copy of attn_fwd_w4i_kernel per compilation (-DW4I_ABL=K; the generated phase
// statements with a class of instructions removed: tools/gen_attn_w4i.py --diag).  Results are WRONG by design — the copies
// exist to price the instruction classes of the hand-ordered stream on hardware (tools/attn_w4i_ablate.py).
#include <math.h>

#define LC_PASTE2(a, b) a##b
#define LC_PASTE(a, b) LC_PASTE2(a, b)
#if !defined(W4I_INC32) || !defined(W4I_INC64) || !defined(W4I_INC96) || !defined(W4I_INC128)
#error "compile with -DW4I_ABL=K -DW4I_INC64=\"attn_w4i_d64_ablK.inc\" -DW4I_INC128=... (leetcuda_amd/build.py build_diag)"
#endif
#if (W4I_ABL & 1)
#define W4I_RING 1
#endif
#define LC_AN_SLOWPATH_SYM LC_PASTE(g_diag_w4i_slowpath_, W4I_ABL)
#define attn_fwd_w4i_kernel LC_PASTE(LC_PASTE(attn_fwd_w4i_abl, W4I_ABL), _kernel)
#define attn_w4i_body LC_PASTE(attn_w4i_body_abl, W4I_ABL)
#define W4I_ONE_SCHED 1
#include "../lc_launch.h"
#include "../attn_w4i.hip"

extern "C" int LC_PASTE(lc_diag_attn_w4i_abl, W4I_ABL)(const void* Q, const void* K, const void* V, void* O, int B, int H, int N, int D,
                                                      void* stream) {
  using namespace lc;
  if (N % 256 != 0 || (D != 64 && D != 128)) return LC_ERR_SHAPE;
  const int nqb = N / 256;
  const dim3 grid((unsigned)((size_t)nqb * B * H)), block(256);
  const float sl2 = (1.0f / sqrtf((float)D)) * 1.4426950408889634f;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (D == 64) {
    auto kern = attn_fwd_w4i_kernel<64, 0>;
    if (int rc = set_dyn_lds(kern, W4G<64>::LDS)) return rc;
    hipLaunchKernelGGL(kern, grid, block, W4G<64>::LDS, st, (const half_t*)Q, (const half_t*)K, (const half_t*)V, (half_t*)O, N, nqb, sl2);
  } else {
    auto kern = attn_fwd_w4i_kernel<128, 0>;
    if (int rc = set_dyn_lds(kern, W4G<128>::LDS)) return rc;
    hipLaunchKernelGGL(kern, grid, block, W4G<128>::LDS, st, (const half_t*)Q, (const half_t*)K, (const half_t*)V, (half_t*)O, N, nqb, sl2);
  }
  return check_launch();
}
