// hgemm_edge.hip — the vectorised edge kernel (late round 6): any M and N, K % 8 == 0 (NN: N % 8 == 0 too), 16-byte aligned pointers.
// What LC_HGEMM_AUTO runs where no tiled kernel divides the shape (until then hgemm_generic_kernel's element-wise staging, 65 - 75 TFLOP/s
// at 2880^3 NN / 8192 x 8256 x 4096 NN; the reference's kernels are not legal on such shapes at all, hgemm_mma_stage.cu:675-676).  Since
// hgemm_mid_edge_kernel (hgemm_mid.hip EDGE; LC_HGEMM_RAGGED) took the shapes with K % 32 == 0, this kernel serves K % 32 != 0 / K < 64.
// 128 x 128 x 64 workgroup tile, 4 wave64 as 2 x 2, wave tile 64 x 64 = 4 x 4 blocks of v_mfma_f32_16x16x32_f16 (operands swapped as
// everywhere: a lane owns 4 consecutive n of one output row).  Global -> registers as 16-byte chunks one K tile ahead (rows / chunks outside the
// matrix read as zeros), registers -> the other of two LDS buffers behind the MFMAs of the current tile: one barrier per tile.  A (and B as [N][K]): 144-byte LDS rows (conflict-free ds_read_b128).  B as [K][N]: the tile stays k-major in LDS
// (256-byte rows of 128 n, 16-byte chunk c of k row r at slot c ^ 2 f(r), f(r) = (r & 3) | ((r >> 3) & 1) << 2) and a fragment is two
// ds_read_b64_tr_b16 (4 k x 16 n blocks: lane i of a 16-lane group addresses k row i >> 2, n 4 (i & 3) .. + 3 and receives column i).
// Blocks: a 1-D grid over two strips of C — the right strip (all rows, columns Ni .. N) and the bottom strip (rows Mi .. M, columns
// 0 .. Ni); Mi = Ni = 0 is the whole matrix.  Same products as hgemm_generic_kernel in another fp32 order; that kernel stays as
// LC_HGEMM_GENERIC (any K, no alignment) and as the cross-check.
#pragma once
#include "lc_common.h"

namespace lc {

constexpr int EM = 128, EN = 128, EK = 64;
constexpr int ESTR = EK + 8;                       // halves per LDS row of the k-contiguous images
constexpr int EDGE_A_BYTES = EM * ESTR * 2;        // 18 KiB
constexpr int EDGE_B_BYTES = EN * ESTR * 2;        // (the [K][N] image is 64 x 256 B = 16 KiB)
constexpr int EDGE_LDS = 2 * (EDGE_A_BYTES + EDGE_B_BYTES);

LC_DEVINL int edge_key(int kr) { return 2 * ((kr & 3) | (((kr >> 3) & 1) << 2)); }   // chunk XOR of k row kr in the [K][N] image

template <bool B_KN>
__global__ __launch_bounds__(256, 2) void hgemm_edge_kernel(const half_t* __restrict__ A, const half_t* __restrict__ B, half_t* __restrict__ C,
                                                          int M, int N, int K, int Mi, int Ni, int nright, int nrc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = wave_id();
  const int wr = wave >> 1, wc = wave & 1;
  const int i = lane & 15, g = lane >> 4;
  const half8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  // block -> strip and tile origin
  int m0, n0;
  {
    const int id = (int)blockIdx.x;
    if (id < nright) {
      const int bm = id / nrc;
      m0 = bm * EM;
      n0 = Ni + (id - bm * nrc) * EN;
    } else {
      const int nbc = Ni / EN, r = id - nright, bm = r / nbc;
      m0 = Mi + bm * EM;
      n0 = (r - bm * nbc) * EN;
    }
  }
  const int n_end = (int)blockIdx.x < nright ? N : Ni;   // (a bottom-strip block never reaches into the right strip: Ni % 128 == 0)

  // chunk c = tid + 256 e (e = 0 .. 3).  A (and B as [N][K]): row (tid >> 3) + 32 e, k chunk tid & 7.  B as [K][N]: k row (tid >> 4) + 16 e,
  // n chunk tid & 15.  Every load is issued from a clamped (always valid) address and zeroed afterwards when its chunk lies outside the
  // matrix: no branch per chunk, the loads of a tile leave back to back.
  struct Regs { half8_t a[4], b[4]; };
  const int ckc = (tid & 7) * 8;                                   // k offset of this thread's chunks in the k-contiguous images
  const half_t* pa[4];
  const half_t* pb[4];
  bool oka[4], okb[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int row = (tid >> 3) + 32 * e;
    oka[e] = m0 + row < M;
    pa[e] = A + (size_t)min(m0 + row, M - 1) * K;
    if constexpr (!B_KN) {
      okb[e] = n0 + row < n_end;
      pb[e] = B + (size_t)min(n0 + row, N - 1) * K;
    }
  }
  const int bnc = n0 + (tid & 15) * 8;                             // [K][N]: this thread's n chunk
  const bool bnok = bnc < n_end;
  const half_t* const pbn = B + min(bnc, N - 8);
  auto fetch = [&](Regs& r, int k0) {
    const int kk = k0 + ckc;
    const bool kok = kk < K;
    const int kcl = min(kk, K - 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const half8_t va = *(const half8_t*)(pa[e] + kcl);
      r.a[e] = (oka[e] && kok) ? va : zero8;
      if constexpr (!B_KN) {
        const half8_t vb = *(const half8_t*)(pb[e] + kcl);
        r.b[e] = (okb[e] && kok) ? vb : zero8;
      } else {
        const int kr = k0 + (tid >> 4) + 16 * e;
        const half8_t vb = *(const half8_t*)(pbn + (size_t)min(kr, K - 1) * N);
        r.b[e] = (bnok && kr < K) ? vb : zero8;
      }
    }
  };
  auto stage = [&](const Regs& r, int buf) {
    char* as = smem + buf * (EDGE_A_BYTES + EDGE_B_BYTES);
    char* bs = as + EDGE_A_BYTES;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = tid + 256 * e;
      *(half8_t*)(as + ((c >> 3) * ESTR + (c & 7) * 8) * 2) = r.a[e];
      if constexpr (!B_KN) {
        *(half8_t*)(bs + ((c >> 3) * ESTR + (c & 7) * 8) * 2) = r.b[e];
      } else {
        const int kr = c >> 4, nc = c & 15;
        *(half8_t*)(bs + kr * 256 + ((nc ^ edge_key(kr)) * 16)) = r.b[e];
      }
    }
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int buf) {
    const char* as = smem + buf * (EDGE_A_BYTES + EDGE_B_BYTES);
    const char* bs = as + EDGE_A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      half8_t af[4], bf[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) af[a] = *(const half8_t*)(as + ((wr * 64 + a * 16 + i) * ESTR + ks * 32 + g * 8) * 2);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if constexpr (!B_KN) {
          bf[b] = *(const half8_t*)(bs + ((wc * 64 + b * 16 + i) * ESTR + ks * 32 + g * 8) * 2);
        } else {
          const int r0 = ks * 32 + g * 8 + (i >> 2), c0 = wc * 8 + 2 * b + ((i & 3) >> 1);
          const half4_t lo = lds_tr16(bs + r0 * 256 + ((c0 ^ edge_key(r0)) * 16) + (i & 1) * 8);
          const half4_t hi = lds_tr16(bs + (r0 + 4) * 256 + ((c0 ^ edge_key(r0 + 4)) * 16) + (i & 1) * 8);
          bf[b] = cat4(lo, hi);
        }
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = mfma16(bf[b], af[a], acc[a][b]);   // swapped: D[n][m]
    }
  };

  // tile t in LDS buffer t & 1, tile t + 1 in flight in registers (requested right behind the staging of tile t, staged behind the MFMAs of tile t)
  const int KT = (K + EK - 1) / EK;
  Regs r;
  fetch(r, 0);
  stage(r, 0);
  fetch(r, EK);
  __syncthreads();
  for (int t = 0; t < KT; t += 2) {
    compute(0);
    if (t + 1 >= KT) break;
    stage(r, 1);
    fetch(r, (t + 2) * EK);
    __syncthreads();
    compute(1);
    if (t + 2 >= KT) break;
    stage(r, 0);
    fetch(r, (t + 3) * EK);
    __syncthreads();
  }
  const bool n4 = (N & 3) == 0;   // 8-byte stores need 8-byte aligned rows
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int gm = m0 + wr * 64 + a * 16 + i;
    if (gm >= M) continue;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int gn = n0 + wc * 64 + b * 16 + g * 4;
      const f32x4_t v = acc[a][b];
      if (n4 && gn + 3 < n_end) {
        half4_t h;
        h[0] = (half_t)v[0]; h[1] = (half_t)v[1]; h[2] = (half_t)v[2]; h[3] = (half_t)v[3];
        *(half4_t*)(C + (size_t)gm * N + gn) = h;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (gn + r < n_end) C[(size_t)gm * N + gn + r] = (half_t)v[r];
      }
    }
  }
}

// dst[r][c] (dst_rows x dst_cols) = r < src_rows && c < src_cols ? src[r][c] : 0, in 16-byte chunks (src_cols, dst_cols % 8 == 0): the zero-padded operand
// copies of LC_HGEMM_KPAD (lc_abi.hip: K % 32 != 0 on a large problem — the padded K runs the tuned kernels, zeros add nothing to a sum)
static __global__ __launch_bounds__(256) void hgemm_pad_copy_kernel(const half_t* __restrict__ src, half_t* __restrict__ dst, int src_rows, int src_cols,
                                                                     int dst_rows, int dst_cols) {
  const size_t cpr = (size_t)dst_cols / 8, i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)dst_rows * cpr) return;
  const size_t r = i / cpr, c = (i - r * cpr) * 8;
  half8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
  if (r < (size_t)src_rows && c < (size_t)src_cols) v = *(const half8_t*)(src + r * (size_t)src_cols + c);
  *(half8_t*)(dst + r * (size_t)dst_cols + c) = v;
}

}  // namespace lc
