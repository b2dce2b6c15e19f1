// tu_attn_w8g.hip — translation unit of the eight-wave D = 64 attention kernel (attn_w8g.hip) — see lc_launch.h
#include <math.h>

// Two waves per SIMD: a wave may use 256 of the SIMD's 512 registers, and every AGPR named in a clobber list counts towards the
// kernel's allocation.  The kernels of this unit own a[0:79]; their asm statements clobber a[0:95] (the allocation granule).
#define LC_AGPR_ALL "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95"
#include "lc_launch.h"
#define LC_AN_SLOWPATH_SYM g_a8_slowpath
#include "attn_w8g.hip"

namespace lc {
// D = 64, N % 256 == 0, V as [B,H,N,D]
int launch_attn_w8g(const half_t* Q, const half_t* K, const half_t* V, half_t* O, int B, int H, int N, int D, hipStream_t st) {
  if (D != 64) return LC_ERR_HEADDIM;
  const int nqb = N / 256;
  const dim3 grid((unsigned)((size_t)nqb * B * H)), block(512);
  const float sl2 = (1.0f / sqrtf((float)D)) * 1.4426950408889634f;
  auto kern = attn_fwd_w8g_kernel<64>;
  if (int rc = set_dyn_lds(kern, W4G<64>::LDS)) return rc;
  hipLaunchKernelGGL(kern, grid, block, W4G<64>::LDS, st, Q, K, V, O, N, nqb, sl2);
  return check_launch();
}
// slow-path counters of THIS unit's kernels, added onto out4[0..2] (out4[3]: last offender, taken when this unit has one)
int diag_attn_slowpath_8(unsigned* out4, int reset) {
  unsigned mine[4] = {0, 0, 0, 0};
  if (hipMemcpyFromSymbol(mine, HIP_SYMBOL(g_a8_slowpath), 16) != hipSuccess) return LC_ERR_LAUNCH;
  if (out4) {
    for (int i = 0; i < 3; ++i) out4[i] += mine[i];
    if (mine[0]) out4[3] = mine[3];
  }
  if (reset) {
    const unsigned z[4] = {0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_a8_slowpath), z, 16) != hipSuccess) return LC_ERR_LAUNCH;
  }
  return LC_OK;
}
}  // namespace lc
