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

// Block -> origin of its 128 x 128 C tile.  rem_base == -1: this kernel's own grid of 128 x 128 tiles.  Otherwise (lc_abi.hip
// launch_mfma256) tiles_m / tiles_n / panel_w describe the 256 x 256 tile grid of the interior and the blocks are, in this order:
//   b < rem_blocks           quadrant b & 3 of the 256-tile whose raster id is rem_base + (b >> 2) — the ids the big kernel's
//                            truncated grid left out (its ragged last wave);
//   the next nright blocks   the right border strip of an N % 256 == 128 problem: columns N - 128 .., 128-row tile b' (all M rows);
//   the rest                 the bottom border strip of an M % 256 == 128 problem: rows M - 128 .., 128-column tile b'' of the
//                            256 tiles_n interior columns (the corner belongs to the right strip).
struct Tile128 {
  int m0, n0;
};
LC_DEVINL Tile128 mfma128_tile(int b, int nblocks, int M, int N, int tiles_m, int tiles_n, int panel_w, int rem_base, int rem_blocks,
                               int nright) {
  Tile128 t;
  if (rem_base == -1) {
    const TileCoord tc = block_tile(b, nblocks, tiles_m, tiles_n, panel_w);
    t.m0 = tc.tm * 128;
    t.n0 = tc.tn * 128;
  } else if (b >= rem_blocks) {
    const int bb = b - rem_blocks;
    if (bb < nright) {
      t.m0 = bb * 128;
      t.n0 = N - 128;
    } else {
      t.m0 = M - 128;
      t.n0 = (bb - nright) * 128;
    }
  } else {
    const int id = rem_base + (b >> 2), qd = b & 3;
    const TileCoord tc = panel_w < 0 ? raster_xcd16(id, tiles_m * tiles_n, tiles_m, tiles_n) : raster(id, tiles_m, tiles_n, panel_w);
    t.m0 = tc.tm * 256 + (qd >> 1) * 128;
    t.n0 = tc.tn * 256 + (qd & 1) * 128;
  }
  return t;
}

// ksplit > 1 (border strips / ragged-tail quadrants only: few tiles with a long K walk, profiles/r5a_hgemm_shapes.log): the grid holds
// ksplit consecutive blocks per tile, block (tile, s) walks the K tiles [s KT / ksplit, (s + 1) KT / ksplit) (the last one also the
// K % 64 == 32 half-step) and writes its fp32 accumulators in the MFMA lane order — 16 coalesced float4 rows per wave — to
// ws[(tile ksplit + s)][wave][16][64]; hgemm_splitk_reduce_kernel adds the ksplit partials and stores C.
// KSW = 2 (round 5): EIGHT waves on the same 128 x 128 tile — waves 4 .. 7 repeat the 2 x 2 wave grid and each group of four takes ONE of
// the two 32-wide k-steps of every K tile (intra-workgroup split-K: half the MFMAs per wave, the same LDS tile, half the DMA pieces per
// wave); at the end group 1 hands its fp32 accumulators over through LDS (4 x 16 KiB = the whole allocation) and group 0 adds and runs
// the epilogue.  Two waves per SIMD inside ONE block: on grids that leave most CUs idle (1024^3: 64 blocks, 1536^3: 144) + 11 %; at one
// block per CU (2048^3) level — the 128 x 128 tile is L2-bandwidth-bound there, not latency-bound — and the NN form loses from 2560^3 on
// (profiles/r5g_hgemm_128w.log), so LC_HGEMM_AUTO uses it up to 0.6 blocks per CU (lc_abi.hip mfma128_ksw).  Never combined with the
// workspace split-K (ksplit must be 1).
template <bool B_KN, int KSW>
__global__ __launch_bounds__(256 * KSW, KSW == 1 ? 2 : 1) void hgemm_mfma128_kernel(const half_t* __restrict__ A,
                                                               const half_t* __restrict__ B,
                                                               half_t* __restrict__ C, int M, int N, int K,
                                                               int tiles_m, int tiles_n, int panel_w, int rem_base,
                                                               int rem_blocks, int nright, int ksplit, float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  static_assert(KSW == 1 || KSW == 2, "KSW: 1 = four waves, 2 = eight waves (one k-step of every K tile per group of four)");
  constexpr int NWAVES = 4 * KSW, NPW = 4 / KSW;   // waves per workgroup, LDS-DMA pieces per wave, operand and K tile
  const int lane = threadIdx.x & 63;
  const int wave = wave_id();
  const int kgrp = wave >> 2;                      // which k-step of a K tile this wave computes (KSW == 2)
  const int wr = (wave >> 1) & 1, wc = wave & 1;
  const int i16 = lane & 15, g = lane >> 4;

  const int tb = ksplit > 1 ? (int)blockIdx.x / ksplit : (int)blockIdx.x, ks = ksplit > 1 ? (int)blockIdx.x % ksplit : 0;
  const Tile128 tl = mfma128_tile(tb, ksplit > 1 ? (int)gridDim.x / ksplit : (int)gridDim.x, M, N, tiles_m, tiles_n, panel_w, rem_base,
                                  rem_blocks, nright);
  const int m0 = tl.m0, n0 = tl.n0;

  // ---- LDS-DMA sources: 16 + 16 pieces of 1 KiB per K tile, NPW + NPW per wave
  const half_t* sa[NPW];
  const half_t* sb[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    // piece i of this wave = 8-row block NWAVES i + wave (K-contiguous operands): the waves' concurrent requests cover 32 / 64 consecutive
    // rows (hgemm_w4y.hip's piece map, docs/HISTORY.md 4.14d); NN B pieces (4 k rows of 256 B) keep one band per wave
    const int p = wave * NPW + i, pk = NWAVES * i + wave;
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
  const int kt0 = ksplit > 1 ? (int)((long)ks * KT / ksplit) : 0, kt1 = ksplit > 1 ? (int)((long)(ks + 1) * KT / ksplit) : KT;
  const size_t bstep = B_KN ? (size_t)BK * N : (size_t)BK;
  auto issue = [&](int t, char* slot) {
#pragma unroll
    for (int i = 0; i < NPW; ++i) glds16(sa[i] + (size_t)t * BK, slot + (NWAVES * i + wave) * 1024);
#pragma unroll
    for (int i = 0; i < NPW; ++i) glds16(sb[i] + (size_t)t * bstep, slot + TILE1_BYTES + (B_KN ? wave * NPW + i : NWAVES * i + wave) * 1024);
  };
  issue(kt0, smem);
  for (int kt = kt0; kt < kt1; ++kt) {
    const char* cur = smem + ((kt - kt0) & 1) * SLOT1_BYTES;
    char* nxt = smem + (((kt - kt0) & 1) ^ 1) * SLOT1_BYTES;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < kt1) issue(kt + 1, nxt);
    const char* la = cur;
    const char* lb = cur + TILE1_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (KSW == 2 && ks != kgrp) continue;   // (wave-uniform: this group's k-step only)
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
  if ((K & 32) && (ksplit <= 1 || ks == ksplit - 1) && kgrp == 0) {   // K % 64 == 32: the last half K-step (split-K: of the last range), fragments straight from global memory in the MFMA operand layout
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
  if constexpr (KSW == 2) {
    // group 1 -> LDS -> group 0: lane-order float4 rows (16 per wave, conflict-free), 16 KiB per wave pair
    __syncthreads();                                  // every wave is done with the last K tile
    f32x4_t* xp = reinterpret_cast<f32x4_t*>(smem) + (wave & 3) * (16 * 64) + lane;
    if (kgrp == 1) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) xp[(mi * 4 + ni) * 64] = acc[mi][ni];
    }
    __syncthreads();
    if (kgrp == 0) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const f32x4_t x = xp[(mi * 4 + ni) * 64];
          acc[mi][ni][0] += x[0]; acc[mi][ni][1] += x[1]; acc[mi][ni][2] += x[2]; acc[mi][ni][3] += x[3];
        }
    }
  }
  if (ksplit > 1) {   // split-K: the fp32 partial of this K range, lane order (hgemm_splitk_reduce_kernel reads it back the same way)
    if (kgrp != 0) return;   // (KSW == 2 is never launched with ksplit > 1)
    f32x4_t* wp = reinterpret_cast<f32x4_t*>(ws) + ((size_t)blockIdx.x * 4 + wave) * (16 * 64) + lane;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) wp[(mi * 4 + ni) * 64] = acc[mi][ni];
    return;
  }
  // ---- epilogue: each wave stages its 64x64 sub-tile through LDS, stores 128-byte row segments
  char* stg = smem + (wave & 3) * (64 * EPI_STRIDE);
  __syncthreads();                                    // (KSW == 2: group 0 has read every partial)
  if (kgrp == 0) {
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
  }
  __syncthreads();
  if (kgrp == 0) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = it * 8 + (lane >> 3);
      const u32x4_t v = *(const u32x4_t*)(stg + row * EPI_STRIDE + (lane & 7) * 16);
      half_t* dst = C + (size_t)(m0 + wr * 64 + row) * N + n0 + wc * 64 + (lane & 7) * 8;
      *(u32x4_t*)dst = v;
    }
  }
}

// Second half of a split-K border launch: block t adds the ksplit fp32 partials of tile t (written by hgemm_mfma128_kernel in MFMA lane
// order) and stores the 128 x 128 fp16 tile: lane (i16, g) of wave (wr, wc) holds C[m0 + 64 wr + 16 mi + i16][n0 + 64 wc + 16 ni + 4 g ..+3].
__global__ __launch_bounds__(256) void hgemm_splitk_reduce_kernel(const float* __restrict__ ws, half_t* __restrict__ C, int M, int N,
                                                                  int tiles_m, int tiles_n, int panel_w, int rem_base, int rem_blocks,
                                                                  int nright, int ksplit) {
  const int lane = threadIdx.x & 63, wave = wave_id();
  const int wr = wave >> 1, wc = wave & 1, i16 = lane & 15, g = lane >> 4;
  const Tile128 tl = mfma128_tile((int)blockIdx.x, (int)gridDim.x, M, N, tiles_m, tiles_n, panel_w, rem_base, rem_blocks, nright);
  const f32x4_t* wp = reinterpret_cast<const f32x4_t*>(ws) + ((size_t)blockIdx.x * ksplit * 4 + wave) * (16 * 64) + lane;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      f32x4_t v = wp[(mi * 4 + ni) * 64];
      for (int s = 1; s < ksplit; ++s) {
        const f32x4_t x = wp[(size_t)s * 4 * (16 * 64) + (mi * 4 + ni) * 64];
        v[0] += x[0]; v[1] += x[1]; v[2] += x[2]; v[3] += x[3];
      }
      half4_t h;
      h[0] = (half_t)v[0]; h[1] = (half_t)v[1]; h[2] = (half_t)v[2]; h[3] = (half_t)v[3];
      *(half4_t*)(C + (size_t)(tl.m0 + wr * 64 + mi * 16 + i16) * N + tl.n0 + wc * 64 + ni * 16 + g * 4) = h;
    }
}

}  // namespace lc
