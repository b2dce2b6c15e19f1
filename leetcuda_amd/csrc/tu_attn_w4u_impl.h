// tu_attn_w4u_impl.h — body of the four translation units tu_attn_w4u_{d128,d128t,d64,d64t}.hip: the merged-phase attention kernel
// (attn_w4u.hip) for ONE (head dim, V layout) and its three block walks.  The includer defines W4U_D, W4U_VT and W4U_TAG.
#include <math.h>

#include <atomic>

#include "lc_launch.h"
#define W4U_CAT2(a, b) a##b
#define W4U_CAT(a, b) W4U_CAT2(a, b)
#define LC_AN_SLOWPATH_SYM W4U_CAT(g_au_slowpath_, W4U_TAG)
#include "attn_w4u.hip"

namespace lc {
namespace {
template <int WALK>
int launch_w4u_walk(const half_t* Q, const half_t* K, const half_t* V, half_t* O, int B, int H, int N, int grid_wgs, size_t nblk,
                    hipStream_t st) {
  constexpr int D = W4U_D;
  static std::atomic<unsigned> ticket{0};     // rotating claim-counter slot of the dynamic walk (attn_w4u.hip g_w4u_queue)
  const int qslot = WALK == 2 ? (int)(ticket.fetch_add(1, std::memory_order_relaxed) % (unsigned)W4U_QSLOTS) : 0;
  const float sl2 = (1.0f / sqrtf((float)D)) * 1.4426950408889634f;
  auto kern = attn_fwd_w4u_kernel<D, W4U_VT, WALK>;
  if (int rc = set_dyn_lds(kern, W4U<D>::LDS)) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid_wgs), dim3(256), W4U<D>::LDS, st, Q, K, V, O, N, N / 256, sl2, (int)nblk, grid_wgs, qslot);
  return check_launch();
}
}  // namespace

// N % 256 == 0; walk: 0 one block per workgroup, 1 persistent static walk, 2 persistent dynamic queue.  A persistent walk with no
// more blocks than CUs IS the one-block launch; the dynamic queue needs a grid that is a multiple of the 8 XCDs.
int W4U_CAT(launch_attn_w4u_, W4U_TAG)(const half_t* Q, const half_t* K, const half_t* V, half_t* O, int B, int H, int N, int walk,
                                       hipStream_t st) {
  const size_t nblk = (size_t)(N / 256) * B * H;
  const int ncu = device_cu_count();   // one workgroup per CU: each takes a CU's whole register file and > half its LDS
  if (walk != 0 && nblk <= (size_t)ncu) walk = 0;
  if (walk == 2 && ncu % 8 != 0) walk = 1;
  if (walk == 0) return launch_w4u_walk<0>(Q, K, V, O, B, H, N, (int)nblk, nblk, st);
  if (walk == 1) return launch_w4u_walk<1>(Q, K, V, O, B, H, N, ncu, nblk, st);
  return launch_w4u_walk<2>(Q, K, V, O, B, H, N, ncu, nblk, st);
}
// slow-path counters of THIS unit's kernels, added onto out4[0..2] (out4[3]: last offender, taken when this unit has one)
int W4U_CAT(diag_attn_slowpath_u_, W4U_TAG)(unsigned* out4, int reset) {
  unsigned mine[4] = {0, 0, 0, 0};
  if (hipMemcpyFromSymbol(mine, HIP_SYMBOL(LC_AN_SLOWPATH_SYM), 16) != hipSuccess) return LC_ERR_LAUNCH;
  if (out4) {
    for (int i = 0; i < 3; ++i) out4[i] += mine[i];
    if (mine[0]) out4[3] = mine[3];
  }
  if (reset) {
    const unsigned z[4] = {0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(LC_AN_SLOWPATH_SYM), z, 16) != hipSuccess) return LC_ERR_LAUNCH;
  }
  return LC_OK;
}
}  // namespace lc
