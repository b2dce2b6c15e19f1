// hgemm_mfma256.hip — fp16 GEMM for gfx950, 256x256x64 workgroup tile, 8 wave64 (2 x 4), LDS-DMA.
//
// Replaces (as a from-scratch CDNA4 design, not a translation) the reference's best NN / TN kernels:
//   kernels/hgemm/mma/basic/hgemm_mma_stage.cu:644-1052   (mma2x4_warp4x4x2_stages_dsmem, NN)
//   kernels/hgemm/mma/swizzle/hgemm_mma_stage_tn_swizzle_x4.cu:207 (TN)
// Semantics: C[M,N] = A[M,K] · B[K,N], fp16 in/out, no alpha/beta. MFMA accumulates in fp32
// (the reference's mma.sync accumulates in fp16, hgemm_mma_stage.cu:110-116) — strictly more accurate.
//
// Data layout in LDS (one 128 KiB dynamic arena, two ring slots of 64 KiB = A tile 32 KiB + B tile 32 KiB):
//   A tile / TN-B tile : [256 rows][64 k] halves, 128 B per row = 8 chunks of 16 B.
//       chunk c of row r is stored at chunk slot  c ^ ((r >> 1) & 7)   ("st_2x8" XOR swizzle):
//       a ds_read_b128 lane group (16 rows x 2 chunks) then covers all 16 16-B slots of a bank row.
//   NN-B tile          : [64 k][256 n] halves, 512 B per row = 16 pairs of 16-B chunks.
//       32-B pair p of row k is stored at pair slot  p ^ h(k),  h(k) = (k & 3) | (((k >> 3) & 1) << 2),
//       read with ds_read_b64_tr_b16 (hardware transpose) — the CDNA4 replacement for ldmatrix.trans.
//   LDS-DMA (global_load_lds_dwordx4) writes lane-linearly, so every swizzle is applied on the per-lane
//   SOURCE address and mirrored on the read address (same involution on both sides).
//
// MFMA: v_mfma_f32_16x16x32_f16 with SWAPPED operands (first = B fragment, second = A fragment) so a
// lane's 4 accumulator registers are 4 consecutive n of one output row -> 8-byte packed epilogue.
#pragma once
#include "lc_common.h"

namespace lc {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;          // 32 KiB (A) == BK * BN * 2 (B)
constexpr int SLOT_BYTES = 2 * TILE_BYTES;       // A + B
constexpr int HGEMM256_LDS = 2 * SLOT_BYTES;     // 128 KiB
constexpr int EPI_STRIDE = 144;                  // bytes per staged C row (64 halves + 16 B pad)

struct TileCoord { int tm, tn; };

// logical tile id -> (tile_m, tile_n): N panels of `panel_w` tiles, M-major inside a panel
// (the reference's thread-block swizzle, hgemm_mma_stage.cu:648 + hgemm.py:198-208, re-expressed).
LC_DEVINL TileCoord raster(int id, int tiles_m, int tiles_n, int panel_w) {
  const int per_panel = panel_w * tiles_m;
  const int panel = id / per_panel;
  const int rem = id - panel * per_panel;
  const int pn0 = panel * panel_w;
  const int w = min(panel_w, tiles_n - pn0);
  TileCoord t;
  t.tm = rem / w;
  t.tn = pn0 + (rem - t.tm * w);
  return t;
}

// ---- XCD super-block raster (round 3; panel_w < 0 selects it, lc_tune_set "hgemm_raster").
// Block b runs on XCD b % 8 (observed; speed only) and blocks start in index order, so at any moment the chip works on ~256
// consecutive block ids = one "step".  The panel raster above gives every XCD its own contiguous id range: good for the
// XCD's L2 (its 32 concurrent tiles are a 4 x 8 block: 12 operand panels), but the 8 XCDs work on regions that share
// nothing — at 16384^3 the step's working set is 8 x 8 B tile-columns + 4 A tile-rows = 544 MiB, twice the 256 MiB Infinity
// Cache, and every B panel is re-streamed from HBM once per step (~1.5 TB/s of HBM traffic ~ 150 W at the 1400 W cap).
// Here a step is ONE compact 16 x 16 block of C tiles (16 + 16 panels = 256 MiB at K = 16384, B panels kept across the steps
// of a 16-column panel) and XCD x owns the 4 x 8 sub-block (x >> 1, x & 1) of it: same 12 panels per XCD from L2's point of
// view, 3x fewer HBM bytes.  Ragged edges (tile grid not a multiple of 16, < 256 trailing blocks) fall back to row-major
// inside the 16-wide panel / identity, so the map is a bijection for every grid (tests/test_layouts.py).
LC_DEVINL TileCoord raster_xcd16(int b, int nwg, int tiles_m, int tiles_n) {
  const int full = nwg & ~255;
  int id = b;
  if (b < full) {
    const int x = b & 7, idx = b >> 3;
    id = ((idx >> 5) << 8) + (x << 5) + (idx & 31);   // step, XCD, slot
  }
  const int per_panel = 16 * tiles_m;
  const int panel = id / per_panel;
  const int rem = id - panel * per_panel;
  const int pn0 = panel * 16;
  const int w = min(16, tiles_n - pn0);
  TileCoord t;
  if (w == 16) {
    const int grp = rem >> 8, r2 = rem & 255;
    if (16 * grp + 16 <= tiles_m) {
      const int sub = r2 >> 5, cc = r2 & 31;
      t.tm = 16 * grp + 4 * (sub >> 1) + (cc >> 3);
      t.tn = pn0 + 8 * (sub & 1) + (cc & 7);
    } else {
      t.tm = 16 * grp + (r2 >> 4);
      t.tn = pn0 + (r2 & 15);
    }
  } else {
    t.tm = rem / w;
    t.tn = pn0 + (rem - t.tm * w);
  }
  return t;
}

// block index -> C tile: the reference's block swizzle (panel_w >= 1: XCD-contiguous ids + N panels) or the XCD super-block raster
LC_DEVINL TileCoord block_tile(int b, int nwg, int tiles_m, int tiles_n, int panel_w) {
  if (panel_w < 0) return raster_xcd16(b, nwg, tiles_m, tiles_n);
  return raster(xcd_remap(b, nwg), tiles_m, tiles_n, panel_w);
}

template <bool B_KN>
struct StageSrc {
  const half_t* a[4];
  const half_t* b[4];
};

// Per-lane global source pointers for the 4 + 4 LDS-DMA pieces this wave stages per K tile.
template <bool B_KN>
LC_DEVINL void stage_src_init(StageSrc<B_KN>& s, const half_t* A, const half_t* B, int m0, int n0,
                              int N, int K, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = wave * 4 + i;                 // piece: 8 rows x 128 B
    const int row = p * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    s.a[i] = A + (size_t)(m0 + row) * K + c * 8;
    if constexpr (!B_KN) {
      s.b[i] = B + (size_t)(n0 + row) * K + c * 8;
    } else {
      const int k = p * 2 + (lane >> 5);        // piece: 2 k-rows x 512 B
      const int pp = lane & 31;
      const int h = (k & 3) | (((k >> 3) & 1) << 2);
      const int nc = (((pp >> 1) ^ h) << 1) | (pp & 1);
      s.b[i] = B + (size_t)k * N + n0 + nc * 8;
    }
  }
}

template <bool B_KN>
LC_DEVINL void stage_issue(const StageSrc<B_KN>& s, char* slot, int wave, size_t koff_a, size_t koff_b) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    glds16(s.a[i] + koff_a, slot + (wave * 4 + i) * 1024);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    glds16(s.b[i] + koff_b, slot + TILE_BYTES + (wave * 4 + i) * 1024);
  }
}

// Lane-dependent LDS read offsets (bytes, relative to the A / B tile of a slot).
template <bool B_KN>
struct FragAddr {
  int a0;        // A fragment, m-tile 0, kstep 0 (kstep 1 = a0 ^ 64; m-tile mi adds mi*2048)
  int b0;        // TN: B fragment, n-tile 0, kstep 0 (same scheme)
  int bt[4];     // NN: tr-read base per n-tile (kstep adds 32*512, second half adds 4*512)
};

template <bool B_KN>
LC_DEVINL void frag_addr_init(FragAddr<B_KN>& f, int wr, int wc, int lane) {
  const int i = lane & 15, g = lane >> 4;
  const int pc0 = g ^ ((lane >> 1) & 7);
  f.a0 = (wr * 128 + i) * 128 + pc0 * 16;
  if constexpr (!B_KN) {
    f.b0 = (wc * 64 + i) * 128 + pc0 * 16;
  } else {
    const int k = 8 * g + (i >> 2);
    const int h = (i >> 2) | ((g & 1) << 2);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      f.bt[ni] = k * 512 + (((wc * 4 + ni) ^ h) * 32) + (i & 3) * 8;
    }
  }
}

template <bool B_KN>
LC_DEVINL void compute_tile(const char* slot, const FragAddr<B_KN>& f, f32x4_t (&acc)[8][4]) {
  const char* la = slot;
  const char* lb = slot + TILE_BYTES;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    half8_t af[8], bf[4];
    half4_t braw[8];   // NN: asm transpose reads (see lc_common.h lds_tr16_asm)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      if constexpr (!B_KN) {
        bf[ni] = *(const half8_t*)(lb + ((f.b0 ^ (ks * 64)) + ni * 2048));
      } else {
        const uint32_t a = lds_addr32(lb + f.bt[ni]) + (uint32_t)(ks * (32 * 512));
        braw[2 * ni] = lds_tr16_asm<0>(a);
        braw[2 * ni + 1] = lds_tr16_asm<4 * 512>(a);
      }
    }
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
      af[mi] = *(const half8_t*)(la + ((f.a0 ^ (ks * 64)) + mi * 2048));
    }
    if constexpr (B_KN) {
      lds_tr16_wait8(braw);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) bf[ni] = cat4(braw[2 * ni], braw[2 * ni + 1]);
    }
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        acc[mi][ni] = mfma16(bf[ni], af[mi], acc[mi][ni]);
      }
    }
  }
}

// Epilogue: fp32 -> fp16, stage each wave's 128x64 sub-tile through LDS in two 64-row passes and
// store whole 128-byte row segments (16 B per lane).
LC_DEVINL void epilogue_store(char* smem, f32x4_t (&acc)[8][4], half_t* C, int N, int m0, int n0,
                              int wave, int wr, int wc, int lane) {
  char* stg = smem + wave * (64 * EPI_STRIDE);
  const int i = lane & 15, g = lane >> 4;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const f32x4_t v = acc[pass * 4 + mi][ni];
        half4_t h;
        h[0] = (half_t)v[0]; h[1] = (half_t)v[1]; h[2] = (half_t)v[2]; h[3] = (half_t)v[3];
        *(half4_t*)(stg + (mi * 16 + i) * EPI_STRIDE + (ni * 16 + g * 4) * 2) = h;
      }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = it * 8 + (lane >> 3);
      const u32x4_t v = *(const u32x4_t*)(stg + row * EPI_STRIDE + (lane & 7) * 16);
      half_t* dst = C + (size_t)(m0 + wr * 128 + pass * 64 + row) * N + n0 + wc * 64 + (lane & 7) * 8;
      *(u32x4_t*)dst = v;
    }
  }
}

// v1 schedule: 2-slot LDS ring filled by LDS-DMA, ONE barrier per K tile; the next tile's DMA is in
// flight during the whole MFMA phase of the current one.
template <bool B_KN>
__global__ __launch_bounds__(512, 2) void hgemm_mfma256_kernel(const half_t* __restrict__ A,
                                                               const half_t* __restrict__ B,
                                                               half_t* __restrict__ C, int M, int N,
                                                               int K, int tiles_m, int tiles_n,
                                                               int panel_w) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = wave_id();
  const int wr = wave >> 2, wc = wave & 3;

  const TileCoord tc = block_tile(blockIdx.x, gridDim.x, tiles_m, tiles_n, panel_w);
  const int m0 = tc.tm * BM, n0 = tc.tn * BN;

  StageSrc<B_KN> src;
  stage_src_init<B_KN>(src, A, B, m0, n0, N, K, wave, lane);
  FragAddr<B_KN> fa;
  frag_addr_init<B_KN>(fa, wr, wc, lane);

  f32x4_t acc[8][4];
#pragma unroll
  for (int mi = 0; mi < 8; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int KT = K / BK;
  const size_t bstep = B_KN ? (size_t)BK * N : (size_t)BK;
  stage_issue<B_KN>(src, smem, wave, 0, 0);
  for (int kt = 0; kt < KT; ++kt) {
    char* cur = smem + (kt & 1) * SLOT_BYTES;
    char* nxt = smem + ((kt & 1) ^ 1) * SLOT_BYTES;
    // tile kt landed (own DMA drained, then the barrier covers every wave's pieces); every wave is
    // also done reading `nxt` (it computed on it in iteration kt-1) before anyone overwrites it.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < KT) stage_issue<B_KN>(src, nxt, wave, (size_t)(kt + 1) * BK, (size_t)(kt + 1) * bstep);
    compute_tile<B_KN>(cur, fa, acc);
  }
  epilogue_store(smem, acc, C, N, m0, n0, wave, wr, wc, lane);
}

}  // namespace lc
