// tu_attn_w4u_impl.h — body of the four translation units tu_attn_w4u_{d128,d128t,d64,d64t}.hip: the merged-phase attention kernel
// (attn_w4u.hip) for ONE (head dim, V layout) and its three block walks.  The includer defines W4U_D, W4U_VT and W4U_TAG.
#include <limits.h>
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
                    hipStream_t st, int nsplit = 1, float* lse = nullptr) {
  constexpr int D = W4U_D;
  static std::atomic<unsigned> ticket{0};     // rotating claim-counter slot of the dynamic walk (attn_w4u.hip g_w4u_queue)
  const int qslot = WALK == 2 ? (int)(ticket.fetch_add(1, std::memory_order_relaxed) % (unsigned)W4U_QSLOTS) : 0;
  const float sl2 = (1.0f / sqrtf((float)D)) * 1.4426950408889634f;
  auto kern = attn_fwd_w4u_kernel<D, W4U_VT, WALK>;
  if (int rc = set_dyn_lds(kern, W4U<D>::LDS)) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid_wgs), dim3(256), W4U<D>::LDS, st, Q, K, V, O, N, (N + 255) / 256, sl2, (int)nblk, grid_wgs, qslot,
                     nsplit, lse);
  return check_launch();
}

// Split-KV (WALK 3): nsplit workgroups per query block write [nsplit][B H][N][D] fp16 partials + [nsplit][B H][N] fp32 log-sum-exps
// into this stream's cached workspace (lc_launch.h stream_workspace), then the combine kernel.  Returns LC_ERR_ARG when the split
// cannot run here (the stream is being captured into a graph — no allocation may happen, and a graph must not keep a pointer into a
// pool that can be regrown — or the allocator refuses): the caller then launches the one-block walk instead, never an error.
int launch_w4u_split(const half_t* Q, const half_t* K, const half_t* V, half_t* O, int B, int H, int N, int nsplit, hipStream_t st) {
  constexpr int D = W4U_D;
  if (stream_is_capturing(st)) return LC_ERR_ARG;
  const size_t rows = (size_t)B * H * N;
  const size_t obytes = (size_t)nsplit * rows * D * sizeof(half_t), lbytes = (size_t)nsplit * rows * sizeof(float);
  WorkspaceLease ws = stream_workspace(st, obytes + lbytes);   // (held until both kernels are enqueued)
  if (!ws.ptr) return LC_ERR_ARG;
  half_t* op = static_cast<half_t*>(ws.ptr);
  float* lse = reinterpret_cast<float*>(static_cast<char*>(ws.ptr) + obytes);
  const size_t nblk = (size_t)(N / 256) * B * H * nsplit;
  if (nblk > (size_t)INT_MAX) return LC_ERR_ARG;   // (the grid and the kernel's block count are ints: the caller runs the unsplit walk)
  if (int rc = launch_w4u_walk<3>(Q, K, V, op, B, H, N, (int)nblk, nblk, st, nsplit, lse)) return rc;
  const size_t threads = rows * (D / 8);
  hipLaunchKernelGGL(attn_split_combine_kernel<D>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, op, lse, O, nsplit, rows);
  return check_launch();
}
}  // namespace

// N % 256 == 0 (walk 0: also N % 256 == 128); walk: 0 one block per workgroup, 1 persistent static walk, 2 persistent dynamic queue.  A persistent walk with no
// more blocks than CUs IS the one-block launch; the dynamic queue needs a grid that is a multiple of the 8 XCDs.
// walk 3 = split-KV with `nsplit` (>= 2, N / 64 % nsplit == 0, >= 2 tiles per split: attn_split_auto, lc_abi.hip) workgroups per query
// block; when the split cannot run on this stream (graph capture, allocator) the one-block walk runs instead.
int W4U_CAT(launch_attn_w4u_, W4U_TAG)(const half_t* Q, const half_t* K, const half_t* V, half_t* O, int B, int H, int N, int walk,
                                       int nsplit, hipStream_t st) {
  const size_t nblk = (size_t)((N + 255) / 256) * B * H;   // (N % 256 != 0: the head's last block is partly real; one block per workgroup only)
  if (nblk > (size_t)INT_MAX) return LC_ERR_SHAPE;           // (int grid / block-count arguments; 2^31 query blocks = 2^39 query rows)
  const int ncu = device_cu_count();   // one workgroup per CU: each takes a CU's whole register file and > half its LDS
  if (N % 256 != 0) walk = 0;          // the persistent walks stage the NEXT block's tiles into ring slots T % 4 == 0 expects; split-KV needs whole blocks
  if (walk == 3) {
    if (nsplit >= 2 && (N / 64) % nsplit == 0 && (N / 64) / nsplit >= 2) {
      const int rc = launch_w4u_split(Q, K, V, O, B, H, N, nsplit, st);
      if (rc != LC_ERR_ARG) return rc;
    }
    walk = 0;
  }
  if (walk != 0 && nblk <= (size_t)ncu) walk = 0;
  if (walk == 2 && ncu % 8 != 0) walk = 1;
  if (walk == 2) {
    // the dynamic queue's claim-counter slot is picked by a host-side ticket AT LAUNCH TIME: captured into a graph it would be baked
    // in, and concurrent replays would share counters (round-4 advisor) -> the static walk while the stream is capturing
    if (stream_is_capturing(st)) walk = 1;
  }
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
