// hgemm_generic.hip — edge-predicated fp16 GEMM for gfx950: any M, N, K (no alignment demands).
// 64x64x32 workgroup tile, 4 wave64 (2 x 2), each wave 32x32 = 2x2 v_mfma_f32_16x16x32_f16 tiles.
// Serves the reference's secondary entry families (kernels/hgemm/naive/hgemm.cu:24-656,
// kernels/hgemm/wmma/hgemm_wmma.cu:47-454, kernels/hgemm/mma/basic/hgemm_mma.cu:80,169) and every
// shape the 256x256x64 kernel does not tile. Element-wise, bounds-checked staging (zero fill);
// NN B tiles are transposed while they are written to LDS so both operands are k-contiguous.
#pragma once
#include "lc_common.h"

namespace lc {

constexpr int GM = 64, GN = 64, GK = 32;
constexpr int GSTR = GK + 8;  // halves per LDS row (80 B: 16-B aligned rows, conflict-free b128 reads)

template <bool B_KN>
__global__ __launch_bounds__(256) void hgemm_generic_kernel(const half_t* __restrict__ A,
                                                            const half_t* __restrict__ B,
                                                            half_t* __restrict__ C, int M, int N, int K) {
  __shared__ __attribute__((aligned(16))) half_t As[GM * GSTR];
  __shared__ __attribute__((aligned(16))) half_t Bs[GN * GSTR];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = wave_id();
  const int wr = wave >> 1, wc = wave & 1;
  const int m0 = blockIdx.y * GM, n0 = blockIdx.x * GN;
  const int i = lane & 15, g = lane >> 4;

  f32x4_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < K; k0 += GK) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int idx = tid + e * 256;
      {  // A[m][k], coalesced along k
        const int r = idx >> 5, kk = idx & 31;
        const int gm = m0 + r, gk = k0 + kk;
        As[r * GSTR + kk] = (gm < M && gk < K) ? A[(size_t)gm * K + gk] : (half_t)0.f;
      }
      if constexpr (!B_KN) {  // B stored [N][K]
        const int r = idx >> 5, kk = idx & 31;
        const int gn = n0 + r, gk = k0 + kk;
        Bs[r * GSTR + kk] = (gn < N && gk < K) ? B[(size_t)gn * K + gk] : (half_t)0.f;
      } else {                // B stored [K][N], coalesced along n, transposed into Bs[n][k]
        const int kk = idx >> 6, c = idx & 63;
        const int gn = n0 + c, gk = k0 + kk;
        Bs[c * GSTR + kk] = (gn < N && gk < K) ? B[(size_t)gk * N + gn] : (half_t)0.f;
      }
    }
    __syncthreads();
    half8_t af[2], bf[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) af[a] = *(const half8_t*)&As[(wr * 32 + a * 16 + i) * GSTR + g * 8];
#pragma unroll
    for (int b = 0; b < 2; ++b) bf[b] = *(const half8_t*)&Bs[(wc * 32 + b * 16 + i) * GSTR + g * 8];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = mfma16(bf[b], af[a], acc[a][b]);  // swapped: D[n][m]
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 2; ++a) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int gm = m0 + wr * 32 + a * 16 + i;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gn = n0 + wc * 32 + b * 16 + g * 4 + r;
        if (gm < M && gn < N) C[(size_t)gm * N + gn] = (half_t)acc[a][b][r];
      }
    }
  }
}

}  // namespace lc
