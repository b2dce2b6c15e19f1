// hgemm_edge.hip — the vectorised edge kernel (late round 6): any M and N, K % 8 == 0 (NN: N % 8 == 0 too), 16-byte aligned pointers.
// What LC_HGEMM_AUTO runs where no tiled kernel divides the shape (until then hgemm_generic_kernel's element-wise staging, 65 - 75 TFLOP/s
// at 2880^3 NN / 8192 x 8256 x 4096 NN; the reference's kernels are not legal on such shapes at all, hgemm_mma_stage.cu:675-676).
// 128 x 128 x 32 workgroup tile, 4 wave64 as 2 x 2, wave tile 64 x 64 = 4 x 4 blocks of v_mfma_f32_16x16x32_f16 (operands swapped as
// everywhere: a lane owns 4 consecutive n of one output row).  Global -> registers as 16-byte chunks one K tile ahead (rows / chunks
// outside the matrix read as zeros), registers -> LDS after the MFMAs of the current tile (two barriers per tile), 80-byte LDS rows
// (conflict-free ds_read_b128).  NN: a chunk is 8 consecutive n of one k row, scattered into the k-contiguous image.  Same products as
// hgemm_generic_kernel in another fp32 order; that kernel stays as LC_HGEMM_GENERIC (any K, no alignment) and as the cross-check.
#pragma once
#include "lc_common.h"

namespace lc {

constexpr int EM = 128, EN = 128, EK = 32;
constexpr int ESTR = EK + 8;   // halves per LDS row

template <bool B_KN>
__global__ __launch_bounds__(256, 2) void hgemm_edge_kernel(const half_t* __restrict__ A, const half_t* __restrict__ B, half_t* __restrict__ C,
                                                          int M, int N, int K) {
  __shared__ __attribute__((aligned(16))) half_t As[EM * ESTR];
  __shared__ __attribute__((aligned(16))) half_t Bs[EN * ESTR];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = wave_id();
  const int wr = wave >> 1, wc = wave & 1;
  const int m0 = blockIdx.y * EM, n0 = blockIdx.x * EN;
  const int i = lane & 15, g = lane >> 4;
  const half8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  // chunk c = tid + 256 e (e = 0, 1).  A (and B as [N][K]): row c >> 2, k chunk c & 3.  B as [K][N]: k row c >> 4, n chunk c & 15.
  half8_t ra[2], rb[2];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int c = tid + 256 * e;
      {
        const int r = c >> 2, kc = (c & 3) * 8;
        const bool ok = (m0 + r < M) && (k0 + kc < K);
        ra[e] = ok ? *(const half8_t*)(A + (size_t)(m0 + r) * K + k0 + kc) : zero8;
      }
      if constexpr (!B_KN) {
        const int r = c >> 2, kc = (c & 3) * 8;
        const bool ok = (n0 + r < N) && (k0 + kc < K);
        rb[e] = ok ? *(const half8_t*)(B + (size_t)(n0 + r) * K + k0 + kc) : zero8;
      } else {
        const int kr = c >> 4, nc = (c & 15) * 8;
        const bool ok = (k0 + kr < K) && (n0 + nc < N);   // (N % 8 == 0: a chunk is inside or outside as a whole)
        rb[e] = ok ? *(const half8_t*)(B + (size_t)(k0 + kr) * N + n0 + nc) : zero8;
      }
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int c = tid + 256 * e;
      *(half8_t*)&As[(c >> 2) * ESTR + (c & 3) * 8] = ra[e];
      if constexpr (!B_KN) {
        *(half8_t*)&Bs[(c >> 2) * ESTR + (c & 3) * 8] = rb[e];
      } else {
        const int kr = c >> 4, nc = (c & 15) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) Bs[(nc + j) * ESTR + kr] = rb[e][j];
      }
    }
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  fetch(0);
  stage();
  __syncthreads();
  for (int k0 = 0; k0 < K; k0 += EK) {
    const bool more = k0 + EK < K;
    if (more) fetch(k0 + EK);
    half8_t af[4], bf[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) af[a] = *(const half8_t*)&As[(wr * 64 + a * 16 + i) * ESTR + g * 8];
#pragma unroll
    for (int b = 0; b < 4; ++b) bf[b] = *(const half8_t*)&Bs[(wc * 64 + b * 16 + i) * ESTR + g * 8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = mfma16(bf[b], af[a], acc[a][b]);   // swapped: D[n][m]
    __syncthreads();   // every wave has read this tile
    if (more) {
      stage();
      __syncthreads();
    }
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
      if (n4 && gn + 3 < N) {
        half4_t h;
        h[0] = (half_t)v[0]; h[1] = (half_t)v[1]; h[2] = (half_t)v[2]; h[3] = (half_t)v[3];
        *(half4_t*)(C + (size_t)gm * N + gn) = h;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (gn + r < N) C[(size_t)gm * N + gn + r] = (half_t)v[r];
      }
    }
  }
}

}  // namespace lc
