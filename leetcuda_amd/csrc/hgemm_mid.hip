// hgemm_mid.hip — the mid-size HGEMM kernel (round 6): (64 | 128 | 192) x (128 | 192) x 64 workgroup tile, 4 wave64 as 2 x 2, an NS-slot LDS ring
// (NS = 3: one workgroup per CU with two K tiles in flight; NS = 2: two workgroups per CU), every instruction of the K loop an asm
// statement in hand-written order: the LDS reads of the next k-step and the LDS-DMA of a later K tile ride in the shadow of this
// k-step's MFMAs.  Reference: the same contraction as kernels/hgemm/mma/basic/hgemm_mma_stage.cu:644-1052 (NN) and
// mma/swizzle/hgemm_mma_stage_tn_swizzle_x4.cu:207 (TN); its default sweep (hgemm.py:28-32,419-421: every multiple of 256 up to 12800)
// is where this kernel earns its place.
//
// Why it exists (profiles/r6a_hgemm_sweep.log, r6a_vendor_kernels.log): between n = 1280 and 2816 LC_HGEMM_AUTO ran hgemm_mfma128_kernel —
// 100 ... 484 workgroups of 128 x 128 — and lost up to 14 % to hipBLASLt (TN), whose heuristic picks per size the macro tile that minimises
// the work of the busiest CU (64 x 128 at 1280, 64 x 160 at 1536, 128 x 128 at 1792 / 2048, 128 x 192 at 2304, 128 x 256 at 2560 / 2816:
// 200 ... 256 workgroups on 256 CUs).  A launch that short (12 ... 45 us, 20 ... 44 K tiles) is decided by latency, not by bytes: every
// workgroup streams operands nobody has touched yet (HBM / Infinity-Cache latency on each K tile, not the 250 ... 400 cycles of an L2
// hit), and with one or two waves per SIMD nothing hides an LDS read or a burst of DMA issue either.  So:
//   * tile shapes (TMW, TNW) in units of 64: lc_abi.hip mid_tile_auto picks the one with the least work on the busiest CU;
//   * grids of one round (<= one workgroup per CU) spend the LDS on DEPTH: NS = 3 slots, the DMA of tile t + 3 issued while tile t
//     computes (a DMA has two whole tiles to land where hgemm_mfma128_kernel has one); larger grids keep NS = 2 and two workgroups per CU;
//   * the k-loop is rotated so that the barrier sits in the MIDDLE of a tile, and every MFMA carries its share of the other work:
//       top      [fragments of tile t, k-step 0 are in registers]   MFMA j of k-step 0  +  LDS read j of k-step 1
//       middle   wait: my DMA pieces of tile t + 1, the k-step 1 fragments;  barrier  -> tile t + 1 visible, slot of tile t free
//       bottom   s_mov m0 / MFMA j of k-step 1 / DMA piece j of tile t + NS -> slot t % NS  +  LDS read j of tile t + 1, k-step 0
//     hipcc schedules none of it (asm volatile statements keep their order); accumulators are AGPR tuples tied in place ("+a").
//     (The builtin form of this loop: hipcc un-ties D from C of the MFMAs and repairs the permutation with 116 v_accvgpr moves per K
//     tile, and its s_waitcnt bookkeeping falls back to lgkmcnt(0) in front of each MFMA group.)
// Shapes: M % (64 TMW) == 0, N % (64 TNW) == 0, K % 32 == 0, K >= 64 (K % 64 == 32: the half k-step of hgemm_mfma128.hip), row strides
// below 2^22 elements (32-bit DMA offsets).  NN (B as [K,N]): TNW = 2 (hgemm_mfma128.hip's [64 k][128 n] transpose image) — its 192-wide
// counterpart is the 192 x 128 tile (TMW = 3: 2304^3 NN in one round of 216 workgroups).  The same kernel also computes the 128 x 128
// quadrants of hgemm_w4y_kernel's ragged last round (rem_base >= 0), and lc_probe_mid256 (liblc_diag.so) instantiates it at 256 x 256.  Operands,
// LDS images and swizzles are the ones of hgemm_mfma128.hip (conflict-freedom: tests/test_layouts.py); arithmetic order per output =
// that kernel's (k ascending, one fp32 accumulator), so the two agree bit for bit (GPU test).
#pragma once
#include "hgemm_mfma256.hip"   // BK, block_tile (templates and inline helpers only)

namespace lc {

template <int TMW, int TNW, int NS>
struct Mid {
  static constexpr int TM = 64 * TMW, TN = 64 * TNW;
  static constexpr int A_TILE = TM * BK * 2;          // 8 / 16 KiB
  static constexpr int B_TILE = TN * BK * 2;          // 16 / 24 KiB
  static constexpr int STAGE = A_TILE + B_TILE;
  static constexpr int MI = 2 * TMW, NI = 2 * TNW;    // 16 x 16 MFMA blocks per wave (wave tile 32 TMW x 32 TNW)
  static constexpr int PA = 2 * TMW, PB = 2 * TNW;    // LDS-DMA pieces of 1 KiB per wave and K tile
  static constexpr int PPW = PA + PB;
  static constexpr int EPI_ROW = 64 * TNW + 16;       // bytes per staged C row of a wave tile (32 TNW halves + 16 B pad)
  static constexpr int EPI = 4 * (32 * TMW) * EPI_ROW;
  static constexpr int LDS = NS * STAGE > EPI ? NS * STAGE : EPI;
  static_assert(TMW >= 1 && TMW <= 4 && TNW >= 2 && TNW <= 4 && NS >= 2 && NS <= 3, "64 ... 256 x 128 / 192 / 256 tiles, 2 .. 3 ring slots");
  static_assert(LDS <= 160 * 1024, "ring must fit a CU's LDS");
  static_assert((NS - 1) * PPW <= 63, "vmcnt is a 6-bit counter");
};

// ---- the asm vocabulary of the loop (all volatile: program order = issue order)
template <int OFF>
LC_DEVINL half8_t lds_rd128_asm(uint32_t lds_byte_addr) {   // (waited for by hand: Frag settle())
  half8_t r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(lds_byte_addr), "n"(OFF));
  return r;
}
template <int N>
LC_DEVINL void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// in-place accumulate, the accumulator TIED to an AGPR tuple
LC_DEVINL void mfma16_acc(f32x4_t& c, half8_t a, half8_t b) {
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
// ... with one LDS-DMA piece around it: m0 (LDS destination of the wave's 1 KiB, wave-uniform) is written in front of the MFMA, the
// load (global address = 64-bit SGPR base + 32-bit per-lane offset) issues behind it — the MFMA is the wait state the m0 write needs
LC_DEVINL void mfma16_acc_dma(f32x4_t& c, half8_t a, half8_t b, uint32_t m0_lds, uint32_t voff, const void* sbase) {
  asm volatile("s_mov_b32 m0, %3\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\tglobal_load_lds_dwordx4 %4, %5"
               : "+a"(c)
               : "v"(a), "v"(b), "s"(m0_lds), "v"(voff), "s"(sbase)
               : "m0", "memory");
}
LC_DEVINL void dma_piece(uint32_t m0_lds, uint32_t voff, const void* sbase) {   // prologue: no MFMA to hide behind
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0_lds), "v"(voff), "s"(sbase) : "m0", "memory");
}
// The last asm MFMAs' results are not visible to hipcc's hazard recogniser: 8 passes = 32 cycles + margin before the first v_accvgpr_read.
// The statement NAMES every accumulator ("+a"): a bare s_nop statement is no fence for hipcc's own register reads — round 6 found the
// epilogue's v_accvgpr_read hoisted above it, and one 16 x 16 block per wave read one MFMA early (only on the K % 64 == 32 path, where the
// last MFMA had nothing behind it).
// A register (tuple) named by an EMPTY volatile statement: volatile statements keep their order, so nothing of hipcc's that reads the
// register can be scheduled above the statement in front of this one (the wait / the settle nops), whatever the number of registers.
template <class T>
LC_DEVINL void name_vgpr(T& x) { asm volatile("" : "+v"(x)); }
template <class T>
LC_DEVINL void name_agpr(T& x) { asm volatile("" : "+a"(x)); }
template <int MI, int NI>
LC_DEVINL void mid_acc_settle(f32x4_t (&acc)[MI][NI]) {
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 7" ::: "memory");
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) name_agpr(acc[mi][ni]);
}
LC_DEVINL const char* sgpr_ptr(const void* p) {   // a pointer hipcc can PROVE wave-uniform (an "s" operand otherwise gets a waterfall loop)
  const uint64_t v = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (const char*)(((uint64_t)hi << 32) | lo);
}

// SK (split-K, round 6): the grid holds ks copies of the tile grid, block b = tile b % tiles of K range b / tiles (whole K tiles, split
// evenly; the last range takes the K % 64 == 32 half step); fp32 partials go to part[ks][M][N] and hgemm_mid_reduce_kernel adds them in
// range order and rounds once.  For shapes whose one-round tile grid covers at most half the CUs and whose K is long (1024 x 1024 x 8192).
// (One body, two entry points below: hgemm_mid_kernel<B_KN, TMW, TNW, NS> and hgemm_mid_sk_kernel<B_KN, TMW, NS> — the names rocprofv3 shows
// and lc_hgemm_kernel_name() reports.)
// EDGE (late round 6, hgemm_mid_edge_kernel below): tiles that reach beyond M / N.  The DMA sources of rows >= M (columns >= N) are CLAMPED to
// the last row (the last whole 16-byte chunk) — a row of C depends on its own row of A only, a column on its own column of B, so what those
// lanes compute is never stored and nothing needs zeroing — and the epilogue stores whole 16-byte chunks inside the matrix only (N % 8 == 0).
// The block map is the edge kernel's: two strips of C, the right one (all rows, columns Ni .. N) and the bottom one (rows Mi .. M, columns
// 0 .. Ni); the arguments tiles_m / tiles_n / panel_w / rem_base carry Mi / Ni / blocks of the right strip / its width in blocks.
template <bool B_KN, int TMW, int TNW, int NS, bool SK, bool EDGE = false>
LC_DEVINL void hgemm_mid_body(const half_t* __restrict__ A, const half_t* __restrict__ B, half_t* __restrict__ C, int M, int N, int K, int tiles_m,
                              int tiles_n, int panel_w, int rem_base, float* __restrict__ part, int ks) {
  using G = Mid<TMW, TNW, NS>;
  static_assert(!B_KN || TNW == 2 || TNW == 4, "NN: whole [64 k][128 n] transpose images");
  constexpr int MI = G::MI, NI = G::NI, PA = G::PA, PB = G::PB, TM = G::TM, TN = G::TN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = wave_id();
  const int wr = wave >> 1, wc = wave & 1;
  const int i16 = lane & 15, g = lane >> 4;
  // rem_base < 0: this kernel's own grid of TM x TN tiles.  rem_base >= 0 (tiles that divide 256 x 256: 128 x 128, 64 x 128; lc_abi.hip
  // launch_mfma256): tiles_m / tiles_n / panel_w describe the 256 x 256 tile grid of hgemm_w4y_kernel and block b is sub-tile b % SUBS
  // (row-major inside the 256-tile) of the 256-tile whose raster id is rem_base + b / SUBS — the ids that kernel's truncated grid left out
  // (its ragged last round; hgemm_mfma128.hip mfma128_tile's map).
  int m0, n0;
  int bid = (int)blockIdx.x, nblk = (int)gridDim.x, srange = 0;
  if constexpr (SK) {
    nblk = EDGE ? panel_w : tiles_m * tiles_n;   // (EDGE: the blocks of the right strip = the whole problem)
    srange = __builtin_amdgcn_readfirstlane(bid / nblk);
    bid -= srange * nblk;
  }
  if constexpr (EDGE) {
    const int Mi = tiles_m, Ni = tiles_n, nright = panel_w, nrc = rem_base;
    if (bid < nright) {
      const int bm = bid / nrc;
      m0 = bm * TM;
      n0 = Ni + (bid - bm * nrc) * TN;
    } else {
      const int nbc = Ni / TN, r = bid - nright, bm = r / nbc;
      m0 = Mi + bm * TM;
      n0 = (r - bm * nbc) * TN;
    }
  } else if (rem_base < 0) {
    const TileCoord tc = block_tile(bid, nblk, tiles_m, tiles_n, panel_w);
    m0 = tc.tm * TM;
    n0 = tc.tn * TN;
  } else if constexpr (256 % TM == 0 && 256 % TN == 0) {
    constexpr int SUBN = 256 / TN, SUBS = (256 / TM) * SUBN;
    const int id = rem_base + (int)blockIdx.x / SUBS, sub = (int)blockIdx.x % SUBS;
    const TileCoord tc = panel_w < 0 ? raster_xcd16(id, tiles_m * tiles_n, tiles_m, tiles_n) : raster(id, tiles_m, tiles_n, panel_w);
    m0 = tc.tm * 256 + (sub / SUBN) * TM;
    n0 = tc.tn * 256 + (sub % SUBN) * TN;
  } else {
    return;   // (never launched: the remainder launcher instantiates 64 / 128 x 128 only)
  }

  // ---- LDS-DMA sources: wave-uniform 64-bit bases (advanced per K tile on the scalar unit) + 32-bit per-lane byte offsets.
  // hgemm_mfma128.hip's piece map: piece i of this wave = 8-row block 4 i + wave of a K-contiguous operand.
  const int KT_all = K / BK;
  int kt0 = 0, KT = KT_all;   // this block's K tiles: kt0 .. kt0 + KT - 1
  if constexpr (SK) {
    kt0 = (int)((long)srange * KT_all / ks);
    KT = (int)((long)(srange + 1) * KT_all / ks) - kt0;
  }
  const char* a_src = sgpr_ptr(A + (size_t)m0 * K + (size_t)kt0 * BK);
  const char* b_src = sgpr_ptr(B_KN ? B + n0 + (size_t)kt0 * BK * N : B + (size_t)n0 * K + (size_t)kt0 * BK);
  const uint32_t b_step = B_KN ? (uint32_t)N * (BK * 2) : (uint32_t)(BK * 2);   // bytes between consecutive K tiles of B
  uint32_t va[PA], vb[PB];
  const int mlim = M - 1 - m0, nlim = N - 1 - n0, clim = N - 8 - n0;   // EDGE: the last row of A / of B as [N][K] / the last whole chunk of B as [K][N], relative to the tile
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int row = (4 * i + wave) * 8 + (lane >> 3);
    const int srow = EDGE ? min(row, mlim) : row;   // (the LDS slot and its swizzle stay the logical row's)
    va[i] = ((uint32_t)srow * (uint32_t)K + (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) * 8)) * 2u;
  }
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    if constexpr (!B_KN) {
      const int row = (4 * i + wave) * 8 + (lane >> 3);
      const int srow = EDGE ? min(row, nlim) : row;
      vb[i] = ((uint32_t)srow * (uint32_t)K + (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) * 8)) * 2u;
    } else {   // sub-image i >> 2 = columns 128 (i >> 2) ..: [64 k][128 n], 256-byte rows = 8 pairs of 16-byte chunks, piece = 4 k rows
      const int p = wave * 4 + (i & 3);
      const int k = p * 4 + (lane >> 4), pp = lane & 15;
      const int h = (k & 3) | (((k >> 3) & 1) << 2);
      const int nc = (((pp >> 1) ^ h) << 1) | (pp & 1);
      const int col = (i >> 2) * 128 + nc * 8;
      vb[i] = ((uint32_t)k * (uint32_t)N + (uint32_t)(EDGE ? min(col, clim) : col)) * 2u;
    }
  }
  const uint32_t smem32 = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr32(smem));
  // LDS destination of DMA piece p (0 .. PPW-1: A pieces first) of this wave inside a slot
  auto piece_lds = [&](int p) -> uint32_t {
    if (p < PA) return (uint32_t)((4 * p + wave) * 1024);
    const int i = p - PA;
    return (uint32_t)(G::A_TILE + (B_KN ? (i >> 2) * 16384 + (wave * 4 + (i & 3)) * 1024 : (4 * i + wave) * 1024));
  };
  // ---- fragment read addresses inside slot 0, per k-step (the k-step flips bit 6 of the swizzled offset: not an immediate)
  const int pc0 = g ^ ((lane >> 1) & 7);
  uint32_t a_rd[2], b_rd[2], bt[B_KN ? NI : 1];
  {
    const int a0 = (wr * (TM / 2) + i16) * 128 + pc0 * 16;
    a_rd[0] = smem32 + (uint32_t)a0;
    a_rd[1] = smem32 + (uint32_t)(a0 ^ 64);
  }
  if constexpr (!B_KN) {
    const int b0 = G::A_TILE + (wc * (TN / 2) + i16) * 128 + pc0 * 16;
    b_rd[0] = smem32 + (uint32_t)b0;
    b_rd[1] = smem32 + (uint32_t)(b0 ^ 64);
  } else {
    const int k = 8 * g + (i16 >> 2);
    const int h = (i16 >> 2) | ((g & 1) << 2);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
      bt[ni] = smem32 + (uint32_t)(G::A_TILE + ((wc * NI + ni) >> 3) * 16384 + k * 256 + ((((wc * NI + ni) & 7) ^ h) * 32) + (i16 & 3) * 8);
  }

  struct Frag {
    half8_t a[MI];
    half8_t b[NI];          // TN: the operand as read; NN: assembled by settle()
    half4_t raw[2 * NI];    // NN: transpose reads
  };
  constexpr int NRD = MI + (B_KN ? 2 * NI : NI);   // LDS read instructions per k-step and wave
  constexpr int NMF = MI * NI;                     // MFMAs per k-step and wave
  constexpr int RPS = (2 * NRD + NMF - 1) / NMF;   // reads per MFMA slot: all of them inside the first half of the MFMAs
  // read r (0 .. NRD-1: A fragments first) of k-step KS out of the slot at byte offset slot_off
  auto read_one = [&](Frag& f, uint32_t slot_off, auto ksc, auto rc) {
    constexpr int ks = decltype(ksc)::value, r = decltype(rc)::value;
    if constexpr (r < MI) {
      f.a[r] = lds_rd128_asm<r * 2048>(a_rd[ks] + slot_off);
    } else if constexpr (!B_KN) {
      f.b[r - MI] = lds_rd128_asm<(r - MI) * 2048>(b_rd[ks] + slot_off);
    } else {
      constexpr int ni = (r - MI) >> 1, hf = (r - MI) & 1;
      f.raw[2 * ni + hf] = lds_tr16_asm<ks * (32 * 256) + hf * (4 * 256)>(bt[ni] + slot_off);
    }
  };
  // every read of f has returned (in-order LDS queue: lgkmcnt(0)); the statement names the registers so that nothing of hipcc's that
  // touches them can be scheduled above it
  auto settle = [&](Frag& f) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) name_vgpr(f.a[mi]);
    if constexpr (!B_KN) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) name_vgpr(f.b[ni]);
    } else {
#pragma unroll
      for (int r = 0; r < 2 * NI; ++r) name_vgpr(f.raw[r]);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) f.b[ni] = cat4(f.raw[2 * ni], f.raw[2 * ni + 1]);
    }
  };
  using KS0 = std::integral_constant<int, 0>;
  using KS1 = std::integral_constant<int, 1>;

  f32x4_t acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // One k-step: MFMA j (mi = j / NI, ni = j % NI) of `cur`; behind it, while any are left, RPS reads of (`nxt`, slot nxt_off, k-step
  // NKS); DMA: the first PPW MFMAs each carry one LDS-DMA piece of the K tile at (a_dma, b_dma) into the slot at dma_off.
  auto kstep = [&](const Frag& cur, Frag& nxt, auto has_nxt, uint32_t nxt_off, auto nks, auto with_dma, uint32_t dma_off, const char* a_dma,
                   const char* b_dma) {
    static_assert(NMF >= G::PPW && NMF * RPS >= NRD, "every DMA piece and every read has an MFMA to ride behind");
    static_for<NMF>([&](auto jc) {
      constexpr int j = decltype(jc)::value, mi = j / NI, ni = j % NI;
      if constexpr (decltype(with_dma)::value && j < G::PPW) {
        if constexpr (j < PA) mfma16_acc_dma(acc[mi][ni], cur.b[ni], cur.a[mi], smem32 + dma_off + piece_lds(j), va[j], a_dma);
        else mfma16_acc_dma(acc[mi][ni], cur.b[ni], cur.a[mi], smem32 + dma_off + piece_lds(j), vb[j - PA], b_dma);
      } else {
        mfma16_acc(acc[mi][ni], cur.b[ni], cur.a[mi]);
      }
      if constexpr (decltype(has_nxt)::value) {
        static_for<RPS>([&](auto qc) {
          constexpr int r = j * RPS + decltype(qc)::value;
          if constexpr (r < NRD) read_one(nxt, nxt_off, nks, std::integral_constant<int, r>{});
        });
      }
    });
  };
  using DMA = std::true_type;
  using NODMA = std::false_type;
  using NEXT = std::true_type;
  using NONEXT = std::false_type;

  // ---- prologue: tiles 0 .. NS - 1 requested (every slot), tile 0 landed and visible, its k-step 0 fragments read
  {
    const char* ap = a_src;
    const char* bp = b_src;
#pragma unroll
    for (int t = 0; t < NS; ++t) {
      if (t < KT) {
        static_for<G::PPW>([&](auto pc) {
          constexpr int p = decltype(pc)::value;
          if constexpr (p < PA) dma_piece(smem32 + (uint32_t)(t * G::STAGE) + piece_lds(p), va[p], ap);
          else dma_piece(smem32 + (uint32_t)(t * G::STAGE) + piece_lds(p), vb[p - PA], bp);
        });
      }
      ap = sgpr_ptr(ap + BK * 2);
      bp = sgpr_ptr(bp + b_step);
    }
  }
  const char* a_nxt = a_src + (size_t)NS * (BK * 2);   // sources of tile kt + NS
  const char* b_nxt = b_src + (size_t)NS * b_step;
  {
    const int behind = (KT < NS ? KT : NS) - 1;   // tiles requested behind tile 0
    if (NS >= 3 && behind >= 2) wait_vmcnt<2 * G::PPW>();
    else if (behind >= 1) wait_vmcnt<G::PPW>();
    else wait_vmcnt<0>();
  }
  raw_barrier();
  Frag f0, f1;
  static_for<NRD>([&](auto rc) { read_one(f0, 0u, KS0{}, rc); });
  settle(f0);
  uint32_t cur = 0;   // byte offset of tile kt's slot (kt % NS)
  // ---- main loop: tiles whose bottom half requests tile kt + NS (no branch inside: an if / else around the two forms of the bottom half
  // gives each its own accumulator registers and hipcc reconciles them with v_accvgpr moves every iteration)
  int kt = 0;
  for (; kt + NS < KT; ++kt) {
    kstep(f0, f1, NEXT{}, cur, KS1{}, NODMA{}, 0u, nullptr, nullptr);
    wait_vmcnt<(NS - 2) * G::PPW>();   // tiles kt + 1 .. kt + NS - 1 are in flight: tile kt + 1 has landed
    settle(f1);
    raw_barrier();   // tile kt + 1 is visible to every wave; every wave has read all of tile kt
    const uint32_t nxt = (cur + G::STAGE == NS * G::STAGE) ? 0u : cur + G::STAGE;
    kstep(f1, f0, NEXT{}, nxt, KS0{}, DMA{}, cur, sgpr_ptr(a_nxt), sgpr_ptr(b_nxt));
    settle(f0);
    cur = nxt;
    a_nxt += BK * 2;
    b_nxt += b_step;
  }
  // ---- the last min(NS, KT) - 1 tiles before the final one: nothing left to request
  for (; kt < KT - 1; ++kt) {
    kstep(f0, f1, NEXT{}, cur, KS1{}, NODMA{}, 0u, nullptr, nullptr);
    if (NS >= 3 && KT - 2 - kt >= 1) wait_vmcnt<G::PPW>();
    else wait_vmcnt<0>();
    settle(f1);
    raw_barrier();
    const uint32_t nxt = (cur + G::STAGE == NS * G::STAGE) ? 0u : cur + G::STAGE;
    kstep(f1, f0, NEXT{}, nxt, KS0{}, NODMA{}, 0u, nullptr, nullptr);
    settle(f0);
    cur = nxt;
  }
  kstep(f0, f1, NEXT{}, cur, KS1{}, NODMA{}, 0u, nullptr, nullptr);
  settle(f1);
  kstep(f1, f0, NONEXT{}, 0u, KS0{}, NODMA{}, 0u, nullptr, nullptr);
  mid_acc_settle<MI, NI>(acc);   // (hipcc may move accumulators between register sets around the branch below: only settled ones)
  if ((__builtin_amdgcn_readfirstlane(K) & 32) && (!SK || srange == ks - 1)) {   // (a scalar branch) K % 64 == 32: the last half K-step, fragments straight from global memory in the MFMA operand layout (hgemm_mfma128.hip)
    const int k0 = KT_all * BK + 8 * g;
    Frag f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int row = wr * (TM / 2) + mi * 16 + i16;
      f.a[mi] = *(const half8_t*)(A + (size_t)(m0 + (EDGE ? min(row, mlim) : row)) * K + k0);
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int col = wc * (TN / 2) + ni * 16 + i16;
      const int scol = EDGE ? min(col, nlim) : col;
      if constexpr (!B_KN) {
        f.b[ni] = *(const half8_t*)(B + (size_t)(n0 + scol) * K + k0);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) f.b[ni][e] = B[(size_t)(k0 + e) * N + n0 + scol];
      }
    }
    // (hipcc assembles the [K][N] fragments with VALU permutes and does not know the asm MFMAs' wait states behind a VALU write: name them, then two wait states)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) name_vgpr(f.a[mi]);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) name_vgpr(f.b[ni]);
    asm volatile("s_nop 1" ::: "memory");
    kstep(f, f0, NONEXT{}, 0u, KS0{}, NODMA{}, 0u, nullptr, nullptr);
    mid_acc_settle<MI, NI>(acc);
  }
  if constexpr (SK) {   // fp32 partials: a lane owns 4 consecutive n of rows mi * 16 + i16 (16-byte stores, 64 B per row and MFMA block)
    // (EDGE: the partials of whole tiles, part[ks][Mp][Np] with Mp x Np = the tile grid's extent — nothing to predicate; the reduce kernel reads what lies inside C)
    const size_t Mp = EDGE ? (size_t)(panel_w / rem_base) * TM : (size_t)M, Np = EDGE ? (size_t)rem_base * TN : (size_t)N;
    float* P = part + ((size_t)srange * Mp + (size_t)(m0 + wr * (TM / 2))) * Np + n0 + wc * (TN / 2);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) *(f32x4_t*)(P + (size_t)(mi * 16 + i16) * Np + ni * 16 + g * 4) = acc[mi][ni];
    return;
  }
  if constexpr (TMW * TNW >= 9) {   // 192 x 192: straight from the accumulators, 8 bytes per lane (the staged epilogue below makes hipcc keep a second
    // copy of the accumulators across the branch above, and 2 x 144 registers exceed the 256 AGPRs)
    half_t* P = C + (size_t)(m0 + wr * (TM / 2)) * N + n0 + wc * (TN / 2);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const f32x4_t v = acc[mi][ni];
        half4_t h;
        h[0] = (half_t)v[0]; h[1] = (half_t)v[1]; h[2] = (half_t)v[2]; h[3] = (half_t)v[3];
        if (!EDGE || (wr * (TM / 2) + mi * 16 + i16 <= mlim && wc * (TN / 2) + ni * 16 + g * 4 <= nlim))   // (N % 8 == 0: 4 columns are inside or outside together)
          *(half4_t*)(P + (size_t)(mi * 16 + i16) * N + ni * 16 + g * 4) = h;
      }
    return;
  }
  // ---- epilogue: each wave stages its (TM / 2) x (TN / 2) sub-tile through LDS, stores whole 16-byte chunks of 64 TNW-byte row segments
  char* stg = smem + wave * ((TM / 2) * G::EPI_ROW);
  __syncthreads();   // every wave is done with the ring
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const f32x4_t v = acc[mi][ni];
      half4_t h;
      h[0] = (half_t)v[0]; h[1] = (half_t)v[1]; h[2] = (half_t)v[2]; h[3] = (half_t)v[3];
      *(half4_t*)(stg + (mi * 16 + i16) * G::EPI_ROW + (ni * 16 + g * 4) * 2) = h;
    }
  }
  __syncthreads();
  constexpr int CPR = 4 * TNW;   // 16-byte chunks per staged row
#pragma unroll
  for (int it = 0; it < CPR * TMW / 2; ++it) {
    const int idx = it * 64 + lane;
    const int row = idx / CPR, c = idx % CPR;
    const u32x4_t v = *(const u32x4_t*)(stg + row * G::EPI_ROW + c * 16);
    if (!EDGE || (wr * (TM / 2) + row <= mlim && wc * (TN / 2) + c * 8 <= clim))
      *(u32x4_t*)(C + (size_t)(m0 + wr * (TM / 2) + row) * N + n0 + wc * (TN / 2) + c * 8) = v;
  }
}

template <bool B_KN, int TMW, int TNW, int NS>
__global__ __launch_bounds__(256, (TMW * TNW >= 6) ? 1 : 2) void hgemm_mid_kernel(const half_t* __restrict__ A, const half_t* __restrict__ B,
                                                           half_t* __restrict__ C, int M, int N, int K, int tiles_m, int tiles_n,
                                                           int panel_w, int rem_base) {
  hgemm_mid_body<B_KN, TMW, TNW, NS, false>(A, B, C, M, N, K, tiles_m, tiles_n, panel_w, rem_base, nullptr, 1);
}
// 128 x 128 tiles that may reach beyond M / N (EDGE above): the border of a ragged shape, or the whole of one (Mi = Ni = 0)
template <bool B_KN, int TMW, int TNW, int NS>
__global__ __launch_bounds__(256, (TMW * TNW >= 6) ? 1 : 2) void hgemm_mid_edge_kernel(const half_t* __restrict__ A, const half_t* __restrict__ B,
                                                                                        half_t* __restrict__ C, int M, int N, int K, int Mi, int Ni, int nright,
                                                                                        int nrc) {
  hgemm_mid_body<B_KN, TMW, TNW, NS, false, true>(A, B, C, M, N, K, Mi, Ni, nright, nrc, nullptr, 1);
}
// ... split-K of a whole ragged problem (64 / 128 x 128 tiles): ks copies of the tile grid, fp32 partials of whole tiles in part[ks][Mp][Np]
template <bool B_KN, int TMW, int NS>
__global__ __launch_bounds__(256, 2) void hgemm_mid_edge_sk_kernel(const half_t* __restrict__ A, const half_t* __restrict__ B, int M, int N, int K, int nright,
                                                                   int nrc, float* __restrict__ part, int ks) {
  hgemm_mid_body<B_KN, TMW, 2, NS, true, true>(A, B, nullptr, M, N, K, 0, 0, nright, nrc, part, ks);
}
// Sum of those partials in range order, rounded once to fp16: 8 elements (one 16-byte chunk of a row of C, N % 8 == 0) per thread.
static __global__ __launch_bounds__(256) void hgemm_mid_reduce_edge_kernel(const float* __restrict__ part, half_t* __restrict__ C, int M, int N, size_t Mp, size_t Np, int ks) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, cpr = (size_t)N / 8;
  if (i >= (size_t)M * cpr) return;
  const size_t row = i / cpr, col = (i - row * cpr) * 8;
  const float* p = part + row * Np + col;
  f32x4_t s0 = *(const f32x4_t*)p, s1 = *(const f32x4_t*)(p + 4);
  for (int r = 1; r < ks; ++r) {
    s0 += *(const f32x4_t*)(p + (size_t)r * Mp * Np);
    s1 += *(const f32x4_t*)(p + (size_t)r * Mp * Np + 4);
  }
  half8_t h;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    h[e] = (half_t)s0[e];
    h[4 + e] = (half_t)s1[e];
  }
  *(half8_t*)(C + row * (size_t)N + col) = h;
}
template <bool B_KN, int TMW, int NS>
__global__ __launch_bounds__(256, 2) void hgemm_mid_sk_kernel(const half_t* __restrict__ A, const half_t* __restrict__ B, int M, int N, int K,
                                                              int tiles_m, int tiles_n, int panel_w, float* __restrict__ part, int ks) {
  hgemm_mid_body<B_KN, TMW, 2, NS, true>(A, B, nullptr, M, N, K, tiles_m, tiles_n, panel_w, -1, part, ks);
}

// Sum of the split-K partials part[ks][mn] in range order, rounded once to fp16: 8 elements per thread.
static __global__ __launch_bounds__(256) void hgemm_mid_reduce_kernel(const float* __restrict__ part, half_t* __restrict__ C, size_t mn, int ks) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i >= mn) return;
  f32x4_t s0 = *(const f32x4_t*)(part + i), s1 = *(const f32x4_t*)(part + i + 4);
  for (int r = 1; r < ks; ++r) {
    s0 += *(const f32x4_t*)(part + (size_t)r * mn + i);
    s1 += *(const f32x4_t*)(part + (size_t)r * mn + i + 4);
  }
  half8_t h;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    h[e] = (half_t)s0[e];
    h[4 + e] = (half_t)s1[e];
  }
  *(half8_t*)(C + i) = h;
}

}  // namespace lc
