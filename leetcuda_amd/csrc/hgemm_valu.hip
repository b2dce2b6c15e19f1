// hgemm_valu.hip — the reference's "CUDA-core" HGEMM ladder (kernels/hgemm/naive/hgemm.cu:24-656,
// kernels/hgemm/naive/hgemm_async.cu) as VECTOR-ALU kernels for gfx950: no MFMA, the same rungs, the same NN contract
// (LC_HGEMM_VALU_*; the 14 `hgemm_naive_f16` / `hgemm_sliced_k_f16` / `hgemm_t_8x8_*` / `hgemm_t_16x8_*` entries).
//
// What a rung means on CDNA4 (one wave64 = 64 lanes, VALU = v_dot2c_f32_f16: two fp16 MACs into an fp32 accumulator per
// lane and instruction — 157 TFLOP/s chip-wide at 2.4 GHz; the reference's rungs accumulate in fp16 with __hfma, this
// ladder keeps fp32 accumulators so that every rung passes the same oracle tolerance as the MFMA kernels):
//   naive      one thread per C element, operands straight from global memory (hgemm.cu:24-39)
//   sliced_k   32 x 32 x 32 block tile through LDS, one C element per thread (hgemm.cu:44-91)
//   t_TMx8     (16 TM) x 128 block tile, 256 threads, TM x 8 register tile per thread, K sliced by BK = 8 / 16 / 32
//              (hgemm.cu:98-806).  LDS holds K-PAIRS: sa[k/2][m] and sb[k/2][n] are half2 {k, k+1}, the operand format of
//              v_dot2c; A pairs are contiguous in global memory, B pairs are packed by the thread that loads rows k, k+1.
//     f16x4 / f16x8   width of the global loads (8 / 16 bytes per lane)
//     _pack           the pair rows are written with vector LDS stores (otherwise dword by dword)
//     _bcf            bank-conflict-free: padded LDS rows (+4 dwords) and a 4 + 4 column split of the thread tile
//                     (columns 4 tx.. and 64 + 4 tx..: a 16-lane group's ds_read_b128 covers all 64 banks once; the plain
//                     rung reads 8 contiguous dwords per lane = 2-way conflicts)
//     _dbuf           double-buffered LDS: the next slice is fetched into registers during the math, one barrier per slice
//     _async          the reference's cp.async rungs: LDS-DMA cannot pack B's k-pairs on the fly, so these entries run the
//                     register-staged _dbuf kernel (documented, not hidden)
// Shapes: M % (16 TM) == 0, N % 128 == 0, K % BK == 0 (the reference's own requirement); anything else takes the
// edge-predicated MFMA kernel (hgemm_generic.hip) like every other family.
#pragma once
#include "lc_common.h"

namespace lc {

LC_DEVINL float valu_dot2(uint32_t a, uint32_t b, float c) {
  return __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, a), __builtin_bit_cast(half2_t, b), c, false);
}

__global__ __launch_bounds__(256) void hgemm_valu_naive_kernel(const half_t* __restrict__ A, const half_t* __restrict__ B,
                                                              half_t* __restrict__ C, int M, int N, int K) {
  const int n = blockIdx.x * 64 + (threadIdx.x & 63), m = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (m >= M || n >= N) return;
  float acc = 0.f;
  for (int k = 0; k < K; ++k) acc += (float)A[(size_t)m * K + k] * (float)B[(size_t)k * N + n];
  C[(size_t)m * N + n] = (half_t)acc;
}

// 32 x 32 x 32 tile, 1024 threads, one C element per thread (M, N, K multiples of 32)
__global__ __launch_bounds__(1024) void hgemm_valu_sliced_k_kernel(const half_t* __restrict__ A, const half_t* __restrict__ B,
                                                                  half_t* __restrict__ C, int M, int N, int K) {
  __shared__ half_t sa[32][32 + 2], sb[32][32 + 2];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int m = blockIdx.y * 32 + ty, n = blockIdx.x * 32 + tx;
  float acc = 0.f;
  for (int k0 = 0; k0 < K; k0 += 32) {
    sa[ty][tx] = A[(size_t)m * K + k0 + tx];
    sb[ty][tx] = B[(size_t)(k0 + ty) * N + n];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 32; ++k) acc += (float)sa[ty][k] * (float)sb[k][tx];
    __syncthreads();
  }
  C[(size_t)m * N + n] = (half_t)acc;
}

// the register-tile rungs.  TM: rows per thread (8 or 16); BKK: K slice; VEC: halves per global load (4 or 8);
// PACK / BCF / DBUF: see the header.
template <int TM, int BKK, int VEC, bool PACK, bool BCF, bool DBUF>
__global__ __launch_bounds__(256) void hgemm_valu_tile_kernel(const half_t* __restrict__ A, const half_t* __restrict__ B,
                                                             half_t* __restrict__ C, int M, int N, int K) {
  constexpr int BM = 16 * TM, BN = 128, KP = BKK / 2, PAD = BCF ? 4 : 0;
  constexpr int SA = BM + PAD, SB = BN + PAD;                    // dwords per pair row
  constexpr int NBUF = DBUF ? 2 : 1;
  __shared__ __attribute__((aligned(16))) uint32_t sa[NBUF][KP][SA];
  __shared__ __attribute__((aligned(16))) uint32_t sb[NBUF][KP][SB];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

  // ---- global -> register staging.  A: chunks of VEC halves along k of one row; B: chunk PAIRS (rows k, k + 1) along n
  constexpr int A_CH = BM * BKK / VEC, A_PER = (A_CH + 255) / 256, A_KC = BKK / VEC;   // chunks, per thread, chunks per row
  constexpr int B_CH = KP * BN / VEC, B_PER = (B_CH + 255) / 256, B_NC = BN / VEC;
  typedef uint32_t chunk_t __attribute__((ext_vector_type(VEC / 2)));                   // VEC halves
  chunk_t ra[A_PER], rb0[B_PER], rb1[B_PER];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int c = tid + 256 * i;
      if (A_CH % 256 == 0 || c < A_CH)
        ra[i] = *(const chunk_t*)(A + (size_t)(m0 + c / A_KC) * K + k0 + (c % A_KC) * VEC);
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const int c = tid + 256 * i;
      if (B_CH % 256 == 0 || c < B_CH) {
        const half_t* p = B + (size_t)(k0 + 2 * (c / B_NC)) * N + n0 + (c % B_NC) * VEC;
        rb0[i] = *(const chunk_t*)p;
        rb1[i] = *(const chunk_t*)(p + N);
      }
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int c = tid + 256 * i;
      if (A_CH % 256 == 0 || c < A_CH) {
        const int m = c / A_KC, kp = (c % A_KC) * (VEC / 2);
#pragma unroll
        for (int e = 0; e < VEC / 2; ++e) sa[buf][kp + e][m] = ra[i][e];               // pair (2 kp + 2e, +1) of row m
      }
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const int c = tid + 256 * i;
      if (B_CH % 256 == 0 || c < B_CH) {
        const int kp = c / B_NC, n = (c % B_NC) * VEC;
        uint32_t w[VEC];   // element e of rows k, k + 1 -> half2 {k, k + 1} of column n + e
#pragma unroll
        for (int e = 0; e < VEC / 2; ++e) {
          w[2 * e] = __builtin_amdgcn_perm(rb1[i][e], rb0[i][e], 0x05040100u);          // low halves of both rows
          w[2 * e + 1] = __builtin_amdgcn_perm(rb1[i][e], rb0[i][e], 0x07060302u);      // high halves
        }
        if constexpr (PACK) {
#pragma unroll
          for (int q = 0; q < VEC / 4; ++q) *(u32x4_t*)&sb[buf][kp][n + 4 * q] = u32x4_t{w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]};
        } else {
#pragma unroll
          for (int e = 0; e < VEC; ++e) sb[buf][kp][n + e] = w[e];
        }
      }
    }
  };

  // ---- thread tile: rows ty-group, columns tx-group.  BCF: 4 + 4 split (conflict-free b128 reads); else 8 contiguous
  float acc[TM][8];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
  auto col_of = [&](int j) { return BCF ? (j < 4 ? 4 * tx + j : 64 + 4 * tx + (j - 4)) : 8 * tx + j; };
  auto row_of = [&](int i) { return BCF ? ((i >> 2) * (BM / (TM / 4)) + 4 * ty + (i & 3)) : TM * ty + i; };
  constexpr int KP_UNROLL = TM == 16 ? 2 : (KP > 8 ? 8 : KP);   // (a fully unrolled 16 x 8 tile over 16 pairs spills)
  auto math = [&](int buf) {
#pragma unroll KP_UNROLL
    for (int kp = 0; kp < KP; ++kp) {
      uint32_t af[TM], bf[8];
#pragma unroll
      for (int i = 0; i < TM; i += 4) {
        const u32x4_t v = *(const u32x4_t*)&sa[buf][kp][row_of(i)];
        af[i] = v[0], af[i + 1] = v[1], af[i + 2] = v[2], af[i + 3] = v[3];
      }
#pragma unroll
      for (int j = 0; j < 8; j += 4) {
        const u32x4_t v = *(const u32x4_t*)&sb[buf][kp][col_of(j)];
        bf[j] = v[0], bf[j + 1] = v[1], bf[j + 2] = v[2], bf[j + 3] = v[3];
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = valu_dot2(af[i], bf[j], acc[i][j]);
    }
  };

  if constexpr (DBUF) {
    fetch(0);
    stage(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < K; k0 += BKK) {
      const bool more = k0 + BKK < K;
      if (more) fetch(k0 + BKK);
      math(buf);
      if (more) stage(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  } else {
    for (int k0 = 0; k0 < K; k0 += BKK) {
      fetch(k0);
      stage(0);
      __syncthreads();
      math(0);
      __syncthreads();
    }
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    half_t* crow = C + (size_t)(m0 + row_of(i)) * N + n0;
#pragma unroll
    for (int j = 0; j < 8; j += 4) {
      half4_t h = {(half_t)acc[i][j], (half_t)acc[i][j + 1], (half_t)acc[i][j + 2], (half_t)acc[i][j + 3]};
      *(half4_t*)(crow + col_of(j)) = h;
    }
  }
}

}  // namespace lc
