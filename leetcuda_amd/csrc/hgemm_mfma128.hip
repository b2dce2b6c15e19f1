// hgemm_mfma128.hip — 128x128x64 workgroup tile, 4 wave64 (2 x 2, wave tile 64x64), LDS-DMA double buffer.
// The mid-size sibling of hgemm_mfma256.hip (same LDS images, swizzles and one-barrier-per-K-tile schedule)
// for the shapes the reference itself accepts but a 256 tile does not divide: M, N multiples of 128, K of 32
// (reference tiles are 128x128x32, kernels/hgemm/mma/basic/hgemm_mma_stage.cu:644-676; K % 64 == 32: one half K-step with
// fragments straight from global memory after the tile loop).  64 KiB of LDS, two workgroups per CU.  Besides its own grid it
// computes, for the 256-tile kernels (lc_abi.hip launch_mfma256), the quadrants of their ragged last wave and the 128-wide border
// strips of M, N % 256 == 128 problems.
#pragma once
#include "hgemm_mfma256.hip"

namespace lc {

constexpr int BM1 = 128, BN1 = 128;
constexpr int TILE1_BYTES = BM1 * BK * 2;        // 16 KiB (A) == BK * BN1 * 2 (B)
constexpr int SLOT1_BYTES = 2 * TILE1_BYTES;
constexpr int HGEMM128_LDS = 2 * SLOT1_BYTES;    // 64 KiB

template <bool B_KN>
__global__ __launch_bounds__(256, 2) void hgemm_mfma128_kernel(const half_t* __restrict__ A,
                                                               const half_t* __restrict__ B,
                                                               half_t* __restrict__ C, int M, int N, int K,
                                                               int tiles_m, int tiles_n, int panel_w, int rem_base,
                                                               int rem_blocks, int nright) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = wave_id();
  const int wr = wave >> 1, wc = wave & 1;
  const int i16 = lane & 15, g = lane >> 4;

  // rem_base == -1: this kernel's own grid of 128 x 128 tiles.  Otherwise (lc_abi.hip launch_mfma256) tiles_m / tiles_n / panel_w
  // describe the 256 x 256 tile grid of the interior and the blocks are, in this order:
  //   b < rem_blocks           quadrant b & 3 of the 256-tile whose raster id is rem_base + (b >> 2) — the ids the big kernel's
  //                            truncated grid left out (its ragged last wave);
  //   the next nright blocks   the right border strip of an N % 256 == 128 problem: columns N - 128 .., 128-row tile b' (all M rows);
  //   the rest                 the bottom border strip of an M % 256 == 128 problem: rows M - 128 .., 128-column tile b'' of the
  //                            256 tiles_n interior columns (the corner belongs to the right strip).
  int m0, n0;
  if (rem_base == -1) {
    const TileCoord tc = block_tile(blockIdx.x, gridDim.x, tiles_m, tiles_n, panel_w);
    m0 = tc.tm * BM1;
    n0 = tc.tn * BN1;
  } else if ((int)blockIdx.x >= rem_blocks) {
    const int bb = (int)blockIdx.x - rem_blocks;
    if (bb < nright) {
      m0 = bb * BM1;
      n0 = N - BN1;
    } else {
      m0 = M - BM1;
      n0 = (bb - nright) * BN1;
    }
  } else {
    const int id = rem_base + ((int)blockIdx.x >> 2), qd = blockIdx.x & 3;
    const TileCoord tc = panel_w < 0 ? raster_xcd16(id, tiles_m * tiles_n, tiles_m, tiles_n) : raster(id, tiles_m, tiles_n, panel_w);
    m0 = tc.tm * 256 + (qd >> 1) * BM1;
    n0 = tc.tn * 256 + (qd & 1) * BN1;
  }

  // ---- LDS-DMA sources: 16 + 16 pieces of 1 KiB per K tile, 4 + 4 per wave
  const half_t* sa[4];
  const half_t* sb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // piece i of this wave = 8-row block 4 i + wave (K-contiguous operands): the four waves' concurrent requests cover 32 consecutive
    // rows (hgemm_w4y.hip's piece map); NN B pieces (4 k rows of 256 B) keep one band per wave
    const int p = wave * 4 + i, pk = 4 * i + wave;
    const int row = pk * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    sa[i] = A + (size_t)(m0 + row) * K + c * 8;
    if constexpr (!B_KN) {
      sb[i] = B + (size_t)(n0 + row) * K + c * 8;
    } else {   // [64 k][128 n] image, 256-byte rows = 8 pairs of 16-byte chunks, piece = 4 k rows
      const int k = p * 4 + (lane >> 4);
      const int pp = lane & 15;
      const int h = (k & 3) | (((k >> 3) & 1) << 2);
      const int nc = (((pp >> 1) ^ h) << 1) | (pp & 1);
      sb[i] = B + (size_t)k * N + n0 + nc * 8;
    }
  }
  // ---- fragment read offsets
  const int pc0 = g ^ ((lane >> 1) & 7);
  const int a0 = (wr * 64 + i16) * 128 + pc0 * 16;
  int b0 = 0, bt[4];
  if constexpr (!B_KN) {
    b0 = (wc * 64 + i16) * 128 + pc0 * 16;
  } else {
    const int k = 8 * g + (i16 >> 2);
    const int h = (i16 >> 2) | ((g & 1) << 2);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) bt[ni] = k * 256 + (((wc * 4 + ni) ^ h) * 32) + (i16 & 3) * 8;
  }

  f32x4_t acc[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int KT = K / BK;
  const size_t bstep = B_KN ? (size_t)BK * N : (size_t)BK;
  auto issue = [&](int t, char* slot) {
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(sa[i] + (size_t)t * BK, slot + (4 * i + wave) * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(sb[i] + (size_t)t * bstep, slot + TILE1_BYTES + (B_KN ? wave * 4 + i : 4 * i + wave) * 1024);
  };
  issue(0, smem);
  for (int kt = 0; kt < KT; ++kt) {
    const char* cur = smem + (kt & 1) * SLOT1_BYTES;
    char* nxt = smem + ((kt & 1) ^ 1) * SLOT1_BYTES;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < KT) issue(kt + 1, nxt);
    const char* la = cur;
    const char* lb = cur + TILE1_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      half8_t af[4], bf[4];
      half4_t braw[8];   // NN: asm transpose reads (the builtin form is guarded by s_waitcnt vmcnt(0) after an LDS-DMA,
                         // which here would serialise the DMA of tile kt+1 with the MFMAs of tile kt)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        if constexpr (!B_KN) {
          bf[ni] = *(const half8_t*)(lb + ((b0 ^ (ks * 64)) + ni * 2048));
        } else {
          const uint32_t a = lds_addr32(lb + bt[ni]) + (uint32_t)(ks * (32 * 256));
          braw[2 * ni] = lds_tr16_asm<0>(a);
          braw[2 * ni + 1] = lds_tr16_asm<4 * 256>(a);
        }
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) af[mi] = *(const half8_t*)(la + ((a0 ^ (ks * 64)) + mi * 2048));
      if constexpr (B_KN) {
        lds_tr16_wait8(braw);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) bf[ni] = cat4(braw[2 * ni], braw[2 * ni + 1]);
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma16(bf[ni], af[mi], acc[mi][ni]);
    }
  }
  if (K & 32) {   // K % 64 == 32: the last half K-step, fragments straight from global memory in the MFMA operand layout
    const int k0 = KT * BK + 8 * g;
    half8_t af[4], bf[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) af[mi] = *(const half8_t*)(A + (size_t)(m0 + wr * 64 + mi * 16 + i16) * K + k0);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      if constexpr (!B_KN) {
        bf[ni] = *(const half8_t*)(B + (size_t)(n0 + wc * 64 + ni * 16 + i16) * K + k0);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) bf[ni][e] = B[(size_t)(k0 + e) * N + n0 + wc * 64 + ni * 16 + i16];
      }
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma16(bf[ni], af[mi], acc[mi][ni]);
  }
  // ---- epilogue: each wave stages its 64x64 sub-tile through LDS, stores 128-byte row segments
  char* stg = smem + wave * (64 * EPI_STRIDE);
  __syncthreads();
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const f32x4_t v = acc[mi][ni];
      half4_t h;
      h[0] = (half_t)v[0]; h[1] = (half_t)v[1]; h[2] = (half_t)v[2]; h[3] = (half_t)v[3];
      *(half4_t*)(stg + (mi * 16 + i16) * EPI_STRIDE + (ni * 16 + g * 4) * 2) = h;
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int row = it * 8 + (lane >> 3);
    const u32x4_t v = *(const u32x4_t*)(stg + row * EPI_STRIDE + (lane & 7) * 16);
    half_t* dst = C + (size_t)(m0 + wr * 64 + row) * N + n0 + wc * 64 + (lane & 7) * 8;
    *(u32x4_t*)dst = v;
  }
}

}  // namespace lc
