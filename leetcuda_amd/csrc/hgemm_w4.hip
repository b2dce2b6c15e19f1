// hgemm_w4.hip — fp16 GEMM for gfx950, 256x256x64 tile, FOUR wave64 (2 x 2), wave tile 128 x 128.
//
// Same contract, LDS images and swizzles as hgemm_pingpong.hip (reference: kernels/hgemm/mma/basic/
// hgemm_mma_stage.cu:644-1052 NN, kernels/hgemm/mma/swizzle/hgemm_mma_stage_tn_swizzle_x4.cu:207 TN).
// Why a second 256-tile design: the 8-wave kernels run the chip into its POWER limit (80 % MFMA-busy at
// ~1.6 GHz), so throughput is bought with energy per FLOP, not with more busy cycles.  A 128x128 wave tile
// reads (128+128) x 16 B per 16 MFMAs = 0.5 KiB of LDS per MFMA instead of 0.75 KiB (128x64), and one wave
// per SIMD needs no s_setprio ping-pong and only ONE barrier per K tile:
//   * one wave per SIMD (256 threads, 512-entry register file): 4x4 v_mfma_f32_32x32x16_f16
//     accumulators = 256 registers (AGPR half), fragments double-buffered in 64 VGPRs;
//   * the K tile is 4 k-steps of 16 MFMAs; the fragment reads of step s+1 (8 ds_read_b128 / 16 tr reads)
//     are issued in the first half of step s, behind MFMAs of the same wave (software pipeline inside one
//     instruction stream, chunked with sched_barrier);
//   * LDS ring of 2 slots: after step 2 of tile t every read of slot t&1 has been issued ->
//     s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier  (tile t+1 has landed for everybody, slot t&1 is dead) ->
//     step 3 carries the 8 A pieces of tile t+2 into slot t&1 and the step-0 reads of tile t+1; step 0 of
//     tile t+1 carries the 8 B pieces (one LDS-DMA piece per 2 MFMAs: with all 16 in one step the second of two
//     back-to-back DMA issues holds the wave past its MFMA cover — measured 0.89 -> 0.82 ms at 8192^3).
//     A DMA piece is in flight for >= 3 k-steps (>= 1536 MFMA cycles) before it is needed.
//   * NN B image: two sub-images of 128 CONTIGUOUS columns ([64 k][256 B] each, 32-B pairs XOR-ed by (k&3)<<1,
//     read with ds_read_b64_tr_b16): every DMA lane group fetches whole 128-B lines.
// DMA source addresses are (wave-uniform 64-bit base) + (32-bit lane offset): 2 offset VGPRs per operand
// instead of 16 pointers (the swizzle key of a row only depends on the parity of its 8-row block).
#pragma once
#include "hgemm_pingpong.hip"

namespace lc {

constexpr int W4B_LDS = 5 * TILE_BYTES;   // A ring 2 x 32 KiB + B ring 3 x 32 KiB
constexpr int W4_EPI_STRIDE = 272;   // bytes per staged C row (128 halves + 16 B pad)

// The 4x4 accumulator tile lives in a[0:255], addressed LITERALLY: hipcc's allocator cannot hold 256 live
// accumulator registers (as values — builtin or "+a" operands — it emits ~1500 v_accvgpr copies and 250 scratch
// accesses per K tile).  Every statement names all 256 AGPRs as clobbered, so the compiler never parks a value of
// its own there (audit: no v_accvgpr_* outside ASMSTART/ASMEND, private_segment_fixed_size 0).
// Hazards handled here, not by hipcc: accumulate chains (D -> C of the next MFMA on the same registers) need no
// wait states; A/B fragments come from compiler-visible ds_reads (hipcc waits for them); the first
// v_accvgpr_read of the epilogue is fenced by w4_mfma_drain() (>= 12 states after an 8-pass MFMA).
template <int IDX>
LC_DEVINL void w4_mfma(half8_t a, half8_t b) {   // a[16 IDX .. +15] += a x b
  asm volatile("v_mfma_f32_32x32x16_f16 a[%2:%3], %0, %1, a[%2:%3]"
               :: "v"(a), "v"(b), "n"(IDX * 16), "n"(IDX * 16 + 15) : LC_AGPR_ALL);
}
template <int IDX0, int IDX1>
LC_DEVINL void w4_mfma2(half8_t a0, half8_t a1, half8_t b) {   // two MFMAs sharing operand b, one statement (no pad between)
  asm volatile("v_mfma_f32_32x32x16_f16 a[%3:%4], %0, %2, a[%3:%4]\n\tv_mfma_f32_32x32x16_f16 a[%5:%6], %1, %2, a[%5:%6]"
               :: "v"(a0), "v"(a1), "v"(b), "n"(IDX0 * 16), "n"(IDX0 * 16 + 15), "n"(IDX1 * 16), "n"(IDX1 * 16 + 15)
               : LC_AGPR_ALL);
}
template <int R>
LC_DEVINL void w4_acc_zero() { asm volatile("v_accvgpr_write_b32 a[%0], 0" :: "n"(R) : LC_AGPR_ALL); }
template <int R>
LC_DEVINL float w4_acc_read() {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(x) : "n"(R));
  return x;
}
LC_DEVINL void w4_mfma_drain() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }

// ABL (perf diagnosis only, results WRONG when non-zero): 1 = no DMA after the prologue, 2 = no per-tile wait +
// barrier, 4 = no fragment reads in the loop (MFMA issue only).
template <bool B_KN, int ABL = 0>
__global__ __launch_bounds__(256) void hgemm_w4_kernel(const half_t* __restrict__ A,
                                                       const half_t* __restrict__ B,
                                                       half_t* __restrict__ C, int M, int N, int K,
                                                       int tiles_m, int tiles_n, int panel_w) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = wave_id();
  const int wr = wave >> 1, wc = wave & 1;
  const int l32 = lane & 31, hi = lane >> 5;

  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const TileCoord tc = raster(id, tiles_m, tiles_n, panel_w);
  const int m0 = tc.tm * BM, n0 = tc.tn * BN;

  // ---- DMA sources.  A (and TN B): this wave stages the 8-row blocks blk = 8*wave + p, p = 0..7; lane ->
  // row (lane>>3) of the block, chunk slot (lane&7) holding logical chunk slot ^ ((row>>1)&7), and
  // (row>>1)&7 = ((lane>>4)&3) | ((p&1)<<2).
  unsigned a_off[2];
#pragma unroll
  for (int par = 0; par < 2; ++par)
    a_off[par] = (unsigned)(lane >> 3) * (unsigned)K * 2u +
                 (unsigned)(((lane & 7) ^ (((lane >> 4) & 3) | (par << 2))) * 16);
  const char* ua = (const char*)(A + (size_t)(m0 + wave * 64) * K);
  const char* ub;
  unsigned b_off[2];
  if constexpr (!B_KN) {
    ub = (const char*)(B + (size_t)(n0 + wave * 64) * K);
    b_off[0] = a_off[0];
    b_off[1] = a_off[1];
  } else {
    // NN: sub-image h = columns [128h, +128) as [64 k][256 B], piece q = 4 k-rows; this wave stages
    // q = 4*wave + p2 of both sub-images.  lane -> k row (lane>>4), 16-B slot pp = lane&15 holding chunk nc
    // (32-B pairs XOR-ed by (k&3)<<1).
    const int pp = lane & 15;
    const int pair = (pp >> 1) ^ (((lane >> 4) & 3) << 1);
    const int nc = pair * 2 + (pp & 1);
    b_off[0] = (unsigned)(lane >> 4) * (unsigned)N * 2u + (unsigned)(nc * 16);
    b_off[1] = b_off[0];
    ub = (const char*)(B + (size_t)(wave * 16) * N + n0);
  }
  const size_t a_blk = (size_t)8 * K * 2;                        // bytes between consecutive 8-row blocks
  const size_t a_kt = (size_t)BK * 2;                            // bytes per K tile along a row
  const size_t b_kt = B_KN ? (size_t)BK * N * 2 : (size_t)BK * 2;
  const size_t b_q = (size_t)4 * N * 2;                          // NN: bytes between consecutive pieces
  // piece g = 0..15 of K tile t -> ring slot `slot`
  auto piece = [&](int g, int t, char* slot) {
    if (g < 8) {
      glds16(ua + (size_t)g * a_blk + (size_t)t * a_kt + a_off[g & 1], slot + (wave * 8 + g) * 1024);
    } else {
      const int p = g - 8;
      if constexpr (!B_KN) {
        glds16(ub + (size_t)p * a_blk + (size_t)t * b_kt + b_off[p & 1],
               slot + TILE_BYTES + (wave * 8 + p) * 1024);
      } else {
        const int h = p >> 2, p2 = p & 3;
        glds16(ub + (size_t)p2 * b_q + (size_t)(256 * h) + (size_t)t * b_kt + b_off[0],
               slot + TILE_BYTES + h * HALF_BYTES + (wave * 4 + p2) * 1024);
      }
    }
  };

  // ---- fragment read addresses (one VGPR per k-step: no address arithmetic in the loop)
  const int swz = (lane >> 1) & 7;
  int a_ad[4], b_ad[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    a_ad[ks] = (wr * 128 + l32) * 128 + (((2 * ks + hi) ^ swz) * 16);
    if constexpr (!B_KN) b_ad[ks] = TILE_BYTES + (wc * 128 + l32) * 128 + (((2 * ks + hi) ^ swz) * 16);
  }
  if constexpr (B_KN) {
    const int i = lane & 15, gi = (lane >> 4) & 1;
    const int k = 8 * hi + (i >> 2);
#pragma unroll
    for (int j = 0; j < 4; ++j)   // n-block j = columns 128wc + 32j + 16gi.. = pair 2j + gi of sub-image wc
      b_ad[j] = TILE_BYTES + wc * HALF_BYTES + k * 256 + (((2 * j + gi) ^ ((i >> 2) << 1)) * 32) + (i & 3) * 8;
  }
  auto read_a = [&](const char* slot, int ks, int i) -> half8_t {
    return *(const half8_t*)(slot + a_ad[ks] + i * 4096);
  };
  half4_t braw_lo, braw_hi;          // NN: raw halves of the transpose read just issued
  half4_t braw[2][8];                // NN: raw B fragments of fragment buffer 0 / 1
  auto read_b = [&](const char* slot, int ks, int j) -> half8_t {
    if constexpr (!B_KN) {
      return *(const half8_t*)(slot + b_ad[ks] + j * 4096);
    } else {
      // (asm transpose reads: hipcc would put s_waitcnt vmcnt(0) in front of the builtin form after every LDS-DMA)
      const uint32_t a = lds_addr32(slot + b_ad[j]) + (uint32_t)(ks * 4096);
      braw_lo = lds_tr16_asm<0>(a);
      braw_hi = lds_tr16_asm<1024>(a);
      return half8_t{};
    }
  };

  static_for<256>([&](auto r) { w4_acc_zero<decltype(r)::value>(); });   // accumulator (i, j) = a[16(4i+j) ..]

  const int KT = K / BK;
  // prologue: tiles 0 and 1 in flight, tile 0 landed, step-0 fragments of tile 0 in registers
#pragma unroll
  for (int g = 0; g < 16; ++g) piece(g, 0, smem);
  if (KT > 1) {
#pragma unroll
    for (int g = 0; g < 16; ++g) piece(g, 1, smem + SLOT_BYTES);
    LC_VMCNT(16);
  } else {
    LC_VMCNT(0);
  }
  pp_barrier();

  half8_t af[2][4], bf[2][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) af[0][i] = read_a(smem, 0, i);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    bf[0][j] = read_b(smem, 0, j);
    if constexpr (B_KN) {
      braw[0][2 * j] = braw_lo;
      braw[0][2 * j + 1] = braw_hi;
    }
  }

  // one k-step: 16 MFMAs from fragment buffer `cb`; the first 4 chunks carry the 8 fragment reads of the
  // next step into buffer cb^1; every chunk may carry 2 DMA pieces.
  // one k-step: 16 MFMAs from fragment buffer cb; chunks 0..3 carry the 8 fragment reads of the next step into
  // buffer cb^1; with DB >= 0 every chunk carries DMA piece DB + c of tile t2 (DB = 0: A pieces, 8: B pieces).
  auto step = [&](auto cbc, const char* rslot, int rks, auto dbc, int t2, char* wslot) {
    constexpr int cb = decltype(cbc)::value;
    constexpr int DB = decltype(dbc)::value;
    if constexpr (B_KN) {   // the asm reads of this buffer were issued >= 8 MFMAs ago
      lds_tr16_wait8(braw[cb]);
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[cb][j] = cat4(braw[cb][2 * j], braw[cb][2 * j + 1]);
      // hipcc may build the operand tuples with v_mov copies right here: VALU write -> (asm) MFMA operand read
      asm volatile("s_nop 1" : "+v"(bf[cb][0]), "+v"(bf[cb][1]), "+v"(bf[cb][2]), "+v"(bf[cb][3]));
    }
    static_for<8>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      // MFMAs FIRST: hipcc guards the first asm consumer of the previous step's fragments with lgkmcnt(0); ahead
      // of this step's reads that wait is free, behind them it would expose a full LDS round trip per step.
      constexpr int i = c >> 1, j0 = 2 * (c & 1);
      w4_mfma<4 * i + j0>(bf[cb][j0], af[cb][i]);
      w4_mfma<4 * i + j0 + 1>(bf[cb][j0 + 1], af[cb][i]);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (c < 4 && !(ABL & 4)) {
        af[cb ^ 1][c] = read_a(rslot, rks, c);
        bf[cb ^ 1][c] = read_b(rslot, rks, c);
        if constexpr (B_KN) {
          braw[cb ^ 1][2 * c] = braw_lo;
          braw[cb ^ 1][2 * c + 1] = braw_hi;
        }
      }
      if constexpr (DB >= 0 && !(ABL & 1)) piece(DB + c, t2, wslot);
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using NoDma = std::integral_constant<int, -1>;
  using DmaA = std::integral_constant<int, 0>;
  using DmaB = std::integral_constant<int, 8>;

  for (int kt = 0; kt < KT; ++kt) {
    char* cur = smem + (kt & 1) * SLOT_BYTES;
    char* nxt = smem + ((kt & 1) ^ 1) * SLOT_BYTES;
    if (kt > 0 && kt + 1 < KT)
      step(I0{}, cur, 1, DmaB{}, kt + 1, nxt);   // B pieces of tile kt+1 (its A pieces went out in step 3 of kt-1)
    else
      step(I0{}, cur, 1, NoDma{}, 0, nullptr);
    step(I1{}, cur, 2, NoDma{}, 0, nullptr);
    step(I0{}, cur, 3, NoDma{}, 0, nullptr);
    // every read of `cur` is issued; tile kt+1 must have landed before anybody reads `nxt`
    if constexpr (!(ABL & 2)) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      pp_barrier();
    }
    if (kt + 2 < KT)
      step(I1{}, nxt, 0, DmaA{}, kt + 2, cur);
    else
      step(I1{}, nxt, 0, NoDma{}, 0, nullptr);
  }

  // ---- epilogue: lane holds C[m = 32i + l32][n = 32j + 8(r>>2) + 4hi + (r&3)]; each wave stages one
  // 32-row block (32 x 128 halves) at a time in its private LDS area and writes 256-B row segments.
  w4_mfma_drain();
  __syncthreads();
  char* stg = smem + wave * (32 * W4_EPI_STRIDE);
  half_t* cw = C + (size_t)(m0 + wr * 128) * N + n0 + wc * 128;
  static_for<4>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    static_for<16>([&](auto qc) {
      constexpr int j = decltype(qc)::value >> 2, rq = decltype(qc)::value & 3;
      half4_t h;
      h[0] = (half_t)w4_acc_read<16 * (4 * i + j) + 4 * rq + 0>();
      h[1] = (half_t)w4_acc_read<16 * (4 * i + j) + 4 * rq + 1>();
      h[2] = (half_t)w4_acc_read<16 * (4 * i + j) + 4 * rq + 2>();
      h[3] = (half_t)w4_acc_read<16 * (4 * i + j) + 4 * rq + 3>();
      *(half4_t*)(stg + l32 * W4_EPI_STRIDE + (j * 32 + 8 * rq + 4 * hi) * 2) = h;
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = it * 4 + (lane >> 4);
      const u32x4_t v = *(const u32x4_t*)(stg + row * W4_EPI_STRIDE + (lane & 15) * 16);
      *(u32x4_t*)(cw + (size_t)(i * 32 + row) * N + (lane & 15) * 8) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  });
}


// ------------------------------------------------------------------------------------------------
// hgemm_w4b_kernel — hgemm_w4_kernel with a THREE-slot ring for B (A 2 x 32 KiB + B 3 x 32 KiB = 160 KiB, all of
// a CU's LDS): the B pieces of tile t+2 go out during steps 1-2 of tile t (their slot has been dead since the end of
// tile t-1), one per 4 MFMAs, and are in flight for >= 5 k-steps; the A pieces of tile t+2 follow in step 3 behind the
// barrier (3 k-steps of flight) — no DMA piece is issued less than 1500 MFMA cycles before it is needed, and the wait
// in front of the barrier is a counted vmcnt(8) (the 8 B pieces of tile t+2 stay in flight across it).
// issue order  ... | B(t+1): steps 1-2 of t-1 | A(t+1): step 3 of t-1 | B(t+2): steps 1-2 of t | A(t+2): step 3 of t ...
// BUF: the DMA pieces are buffer_load ... lds (descriptor + scalar offset) instead of global_load_lds (64-bit lane address)
// STAMPS (diagnosis only; clobbers the first bytes of A): s_memtime of wave 0 of workgroup 0 at the step boundaries of
// K tiles 32..35 (lc_tune_set "hgemm_stamps", tools/hgemm_stamps.py).
template <bool B_KN, bool BUF = false, bool STAMPS = false>
__global__ __launch_bounds__(256) void hgemm_w4b_kernel(const half_t* __restrict__ A,
                                                       const half_t* __restrict__ B,
                                                       half_t* __restrict__ C, int M, int N, int K,
                                                       int tiles_m, int tiles_n, int panel_w) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = wave_id();
  const int wr = wave >> 1, wc = wave & 1;
  const int l32 = lane & 31, hi = lane >> 5;

  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const TileCoord tc = raster(id, tiles_m, tiles_n, panel_w);
  const int m0 = tc.tm * BM, n0 = tc.tn * BN;

  // ---- DMA sources.  A (and TN B): this wave stages the 8-row blocks blk = 8*wave + p, p = 0..7; lane ->
  // row (lane>>3) of the block, chunk slot (lane&7) holding logical chunk slot ^ ((row>>1)&7), and
  // (row>>1)&7 = ((lane>>4)&3) | ((p&1)<<2).
  unsigned a_off[2];
#pragma unroll
  for (int par = 0; par < 2; ++par)
    a_off[par] = (unsigned)(lane >> 3) * (unsigned)K * 2u +
                 (unsigned)(((lane & 7) ^ (((lane >> 4) & 3) | (par << 2))) * 16);
  const char* ua = (const char*)(A + (size_t)(m0 + wave * 64) * K);
  const char* ub;
  unsigned b_off[2];
  if constexpr (!B_KN) {
    ub = (const char*)(B + (size_t)(n0 + wave * 64) * K);
    b_off[0] = a_off[0];
    b_off[1] = a_off[1];
  } else {
    // NN: sub-image h = columns [128h, +128) as [64 k][256 B], piece q = 4 k-rows; this wave stages
    // q = 4*wave + p2 of both sub-images.  lane -> k row (lane>>4), 16-B slot pp = lane&15 holding chunk nc
    // (32-B pairs XOR-ed by (k&3)<<1).
    const int pp = lane & 15;
    const int pair = (pp >> 1) ^ (((lane >> 4) & 3) << 1);
    const int nc = pair * 2 + (pp & 1);
    b_off[0] = (unsigned)(lane >> 4) * (unsigned)N * 2u + (unsigned)(nc * 16);
    b_off[1] = b_off[0];
    ub = (const char*)(B + (size_t)(wave * 16) * N + n0);
  }
  const size_t a_blk = (size_t)8 * K * 2;                        // bytes between consecutive 8-row blocks
  const size_t a_kt = (size_t)BK * 2;                            // bytes per K tile along a row
  const size_t b_kt = B_KN ? (size_t)BK * N * 2 : (size_t)BK * 2;
  const size_t b_q = (size_t)4 * N * 2;                          // NN: bytes between consecutive pieces
  // piece g = 0..15 of K tile t (clamped: past the end the last tile is staged again into a dead slot, which keeps
  // the vmcnt counts exact) -> `slot` = the A slot (g < 8) or the B slot (g >= 8) of that tile
  const int KT = K / BK;
  const buf_rsrc_t ra = make_rsrc(ua), rb = make_rsrc(ub);
  auto dma = [&](buf_rsrc_t r, const char* base, size_t uoff, unsigned loff, char* dst) {
    if constexpr (BUF) blds16(r, loff, (unsigned)uoff, dst);
    else glds16(base + uoff + loff, dst);
  };
  auto piece = [&](int g, int t, char* slot) {
    const int te = t < KT ? t : KT - 1;
    if (g < 8) {
      dma(ra, ua, (size_t)g * a_blk + (size_t)te * a_kt, a_off[g & 1], slot + (wave * 8 + g) * 1024);
    } else {
      const int p = g - 8;
      if constexpr (!B_KN) {
        dma(rb, ub, (size_t)p * a_blk + (size_t)te * b_kt, b_off[p & 1], slot + (wave * 8 + p) * 1024);
      } else {
        const int h = p >> 2, p2 = p & 3;
        dma(rb, ub, (size_t)p2 * b_q + (size_t)(256 * h) + (size_t)te * b_kt, b_off[0],
            slot + h * HALF_BYTES + (wave * 4 + p2) * 1024);
      }
    }
  };
  auto a_slot = [&](int t) -> char* { return smem + (t & 1) * TILE_BYTES; };
  auto b_slot = [&](int bi) -> char* { return smem + (2 + bi) * TILE_BYTES; };   // bi = t % 3, tracked by the caller

  // ---- fragment read addresses (one VGPR per k-step: no address arithmetic in the loop)
  const int swz = (lane >> 1) & 7;
  int a_ad[4], b_ad[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    a_ad[ks] = (wr * 128 + l32) * 128 + (((2 * ks + hi) ^ swz) * 16);
    if constexpr (!B_KN) b_ad[ks] = (wc * 128 + l32) * 128 + (((2 * ks + hi) ^ swz) * 16);
  }
  if constexpr (B_KN) {
    const int i = lane & 15, gi = (lane >> 4) & 1;
    const int k = 8 * hi + (i >> 2);
#pragma unroll
    for (int j = 0; j < 4; ++j)   // n-block j = columns 128wc + 32j + 16gi.. = pair 2j + gi of sub-image wc
      b_ad[j] = wc * HALF_BYTES + k * 256 + (((2 * j + gi) ^ ((i >> 2) << 1)) * 32) + (i & 3) * 8;
  }
  auto read_a = [&](const char* slot, int ks, int i) -> half8_t {
    return *(const half8_t*)(slot + a_ad[ks] + i * 4096);
  };
  half4_t braw_lo, braw_hi;          // NN: raw halves of the transpose read just issued
  half4_t braw[2][8];                // NN: raw B fragments of fragment buffer 0 / 1
  auto read_b = [&](const char* slot, int ks, int j) -> half8_t {
    if constexpr (!B_KN) {
      return *(const half8_t*)(slot + b_ad[ks] + j * 4096);
    } else {
      // (asm transpose reads: hipcc would put s_waitcnt vmcnt(0) in front of the builtin form after every LDS-DMA)
      const uint32_t a = lds_addr32(slot + b_ad[j]) + (uint32_t)(ks * 4096);
      braw_lo = lds_tr16_asm<0>(a);
      braw_hi = lds_tr16_asm<1024>(a);
      return half8_t{};
    }
  };

  static_for<256>([&](auto r) { w4_acc_zero<decltype(r)::value>(); });   // accumulator (i, j) = a[16(4i+j) ..]

  // prologue: B(0) A(0) B(1) A(1); tile 0 landed, step-0 fragments of tile 0 in registers
#pragma unroll
  for (int g = 8; g < 16; ++g) piece(g, 0, b_slot(0));
#pragma unroll
  for (int g = 0; g < 8; ++g) piece(g, 0, a_slot(0));
#pragma unroll
  for (int g = 8; g < 16; ++g) piece(g, 1, b_slot(1));
#pragma unroll
  for (int g = 0; g < 8; ++g) piece(g, 1, a_slot(1));
  LC_VMCNT(16);
  pp_barrier();

  half8_t af[2][4], bf[2][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) af[0][i] = read_a(a_slot(0), 0, i);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    bf[0][j] = read_b(b_slot(0), 0, j);
    if constexpr (B_KN) {
      braw[0][2 * j] = braw_lo;
      braw[0][2 * j + 1] = braw_hi;
    }
  }

  // one k-step: 16 MFMAs from fragment buffer cb; chunks 0..3 carry the 8 fragment reads of the next step into
  // buffer cb^1; DMA: NP pieces g0.. of tile t2 into wslot, one per 8/NP chunks (NP = 0, 4 or 8)
  auto step = [&](auto cbc, const char* ra, const char* rb, int rks, auto npc, int g0, int t2, char* wslot) {
    constexpr int cb = decltype(cbc)::value;
    constexpr int NP = decltype(npc)::value;
    if constexpr (B_KN) {   // the asm reads of this buffer were issued >= 8 MFMAs ago
      lds_tr16_wait8(braw[cb]);
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[cb][j] = cat4(braw[cb][2 * j], braw[cb][2 * j + 1]);
      asm volatile("s_nop 1" : "+v"(bf[cb][0]), "+v"(bf[cb][1]), "+v"(bf[cb][2]), "+v"(bf[cb][3]));
    }
    static_for<8>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      constexpr int i = c >> 1, j0 = 2 * (c & 1);
      w4_mfma2<4 * i + j0, 4 * i + j0 + 1>(bf[cb][j0], bf[cb][j0 + 1], af[cb][i]);
      __builtin_amdgcn_sched_barrier(0);
      // (one read per chunk over all 8 chunks instead of two in chunks 0..3 measures the same: A/B run r04c)
      if constexpr (c < 4) {
        af[cb ^ 1][c] = read_a(ra, rks, c);
        bf[cb ^ 1][c] = read_b(rb, rks, c);
        if constexpr (B_KN) {
          braw[cb ^ 1][2 * c] = braw_lo;
          braw[cb ^ 1][2 * c + 1] = braw_hi;
        }
      }
      if constexpr (NP == 8) piece(g0 + c, t2, wslot);
      if constexpr (NP == 4 && (c & 1) == 1) piece(g0 + (c >> 1), t2, wslot);
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using P0 = std::integral_constant<int, 0>;
  using P4 = std::integral_constant<int, 4>;
  using P8 = std::integral_constant<int, 8>;

  unsigned long long* stamp = reinterpret_cast<unsigned long long*>(const_cast<half_t*>(A));
  const bool stamping = STAMPS && blockIdx.x == 0 && wave == 0 && lane == 0;
  auto STAMP = [&](int kt, int k) {
    if constexpr (STAMPS) {
      if (kt >= 32 && kt < 36) {
        const unsigned long long c = __builtin_readcyclecounter();
        if (stamping) stamp[(kt - 32) * 8 + k] = c;
      }
    }
  };
  int b0 = 0, b1 = 1, b2 = 2;   // B slot indices of tiles kt, kt+1, kt+2 (rotating, kt % 3)
  for (int kt = 0; kt < KT; ++kt) {
    const char* ca = a_slot(kt);
    const char* cbs = b_slot(b0);
    STAMP(kt, 0);
    step(I0{}, ca, cbs, 1, P0{}, 0, 0, nullptr);
    STAMP(kt, 1);
    step(I1{}, ca, cbs, 2, P4{}, 8, kt + 2, b_slot(b2));
    STAMP(kt, 2);
    step(I0{}, ca, cbs, 3, P4{}, 12, kt + 2, b_slot(b2));
    STAMP(kt, 3);
    // every read of tile kt is issued; A(kt+1), B(kt+1) must have landed (the 8 B pieces of tile kt+2 stay in flight)
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    STAMP(kt, 4);
    pp_barrier();
    STAMP(kt, 5);
    step(I1{}, a_slot(kt + 1), b_slot(b1), 0, P8{}, 0, kt + 2, a_slot(kt));
    STAMP(kt, 6);
    const int t = b0;
    b0 = b1;
    b1 = b2;
    b2 = t;
  }
  LC_VMCNT(0);

  // ---- epilogue: lane holds C[m = 32i + l32][n = 32j + 8(r>>2) + 4hi + (r&3)]; each wave stages one
  // 32-row block (32 x 128 halves) at a time in its private LDS area and writes 256-B row segments.
  w4_mfma_drain();
  __syncthreads();
  char* stg = smem + wave * (32 * W4_EPI_STRIDE);
  half_t* cw = C + (size_t)(m0 + wr * 128) * N + n0 + wc * 128;
  static_for<4>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    static_for<16>([&](auto qc) {
      constexpr int j = decltype(qc)::value >> 2, rq = decltype(qc)::value & 3;
      half4_t h;
      h[0] = (half_t)w4_acc_read<16 * (4 * i + j) + 4 * rq + 0>();
      h[1] = (half_t)w4_acc_read<16 * (4 * i + j) + 4 * rq + 1>();
      h[2] = (half_t)w4_acc_read<16 * (4 * i + j) + 4 * rq + 2>();
      h[3] = (half_t)w4_acc_read<16 * (4 * i + j) + 4 * rq + 3>();
      *(half4_t*)(stg + l32 * W4_EPI_STRIDE + (j * 32 + 8 * rq + 4 * hi) * 2) = h;
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = it * 4 + (lane >> 4);
      const u32x4_t v = *(const u32x4_t*)(stg + row * W4_EPI_STRIDE + (lane & 15) * 16);
      *(u32x4_t*)(cw + (size_t)(i * 32 + row) * N + (lane & 15) * 8) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  });
}


// ------------------------------------------------------------------------------------------------
// hgemm_w4s_kernel — same 4-wave / 128x128 wave tile / literal-AGPR design, but the LDS ring is 4 STAGES of
// 32 k (A [256 rows][64 B] + B 16 KiB = 32 KiB per stage) instead of 2 slots of 64 k.  Measured reason
// (tools/w4_ablate.py, 8192^3 TN): in the 2-slot kernel all 16 LDS-DMA pieces of a K tile must be issued in the
// one k-step after the barrier (WAR on the slot) and cost 0.17 ms of 0.89 (two back-to-back DMA issues between
// MFMAs hold the wave ~60 cycles each while only 32 are covered); spread 8 + 8 it already matches the 8-wave
// ping-pong kernel.  With 32-k stages a slot is recycled every 2 k-steps, so the 8 pieces per stage go out ONE PER
// FOUR MFMAs, stay in flight for >= 4 k-steps (>= 2048 MFMA cycles) behind a counted vmcnt(16), and three stages
// are always ahead of the MFMAs.
//   stage s lives in ring slot s & 3;  per stage:
//     A(s): 16 MFMAs (k-step 0)  | reads fragments (s, k-step 1)      | DMA pieces 4..7 of stage s+3
//           s_waitcnt vmcnt(16) lgkmcnt(0); s_barrier   -> stage s+1 landed everywhere, slot s&3 dead
//     B(s): 16 MFMAs (k-step 1)  | reads fragments (s+1, k-step 0)    | DMA pieces 0..3 of stage s+4 -> slot s&3
//   issue order ... | s+1 | s+2 | s+3 (4 in B(s-1), 4 in A(s)) : 16 pieces are younger than stage s+1 at the wait.
// 64-byte rows: chunk c (16 B) of row r is stored at chunk slot c ^ ((r>>2)&3); a ds_read_b128 lane group
// ({0-3,12-15,20-27}: four row quads with distinct (r>>2)&3) then covers 16 distinct 16-B positions of a
// 256-B bank row.  Stages past the end re-load the last stage into a dead slot (keeps the counts exact).
constexpr int W4S_STAGE = 32768;          // bytes per stage (A 16 KiB + B 16 KiB)
constexpr int W4S_BK = 32;

template <bool B_KN>
__global__ __launch_bounds__(256) void hgemm_w4s_kernel(const half_t* __restrict__ A,
                                                        const half_t* __restrict__ B,
                                                        half_t* __restrict__ C, int M, int N, int K,
                                                        int tiles_m, int tiles_n, int panel_w) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = wave_id();
  const int wr = wave >> 1, wc = wave & 1;
  const int l32 = lane & 31, hi = lane >> 5;

  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const TileCoord tc = raster(id, tiles_m, tiles_n, panel_w);
  const int m0 = tc.tm * BM, n0 = tc.tn * BN;
  const int NS = K / W4S_BK;

  // ---- DMA sources: A (and TN B) piece p = 0..3 of this wave = rows 16(4 wave + p) .. +15; lane -> row
  // (lane>>2), chunk slot (lane&3) holding logical chunk slot ^ ((row>>2)&3) = slot ^ ((lane>>4)&3).
  const unsigned a_off = (unsigned)(lane >> 2) * (unsigned)K * 2u +
                         (unsigned)(((lane & 3) ^ ((lane >> 4) & 3)) * 16);
  const char* ua = (const char*)(A + (size_t)(m0 + wave * 64) * K);
  const char* ub;
  unsigned b_off;
  if constexpr (!B_KN) {
    ub = (const char*)(B + (size_t)(n0 + wave * 64) * K);
    b_off = a_off;
  } else {
    // NN: sub-image h = columns [128h, +128) as [32 k][256 B] (8 KiB), piece q = 4 k-rows; this wave stages
    // q = 2 wave + p2 of both sub-images
    const int pp = lane & 15;
    const int pair = (pp >> 1) ^ (((lane >> 4) & 3) << 1);
    const int nc = pair * 2 + (pp & 1);
    b_off = (unsigned)(lane >> 4) * (unsigned)N * 2u + (unsigned)(nc * 16);
    ub = (const char*)(B + (size_t)(wave * 8) * N + n0);
  }
  const size_t a_p = (size_t)16 * K * 2;                          // bytes between consecutive 16-row pieces
  const size_t b_st = B_KN ? (size_t)W4S_BK * N * 2 : (size_t)W4S_BK * 2;
  const size_t b_q = (size_t)4 * N * 2;
  // piece g = 0..7 of stage st (clamped) -> ring slot st & 3
  auto piece = [&](int g, int st) {
    const int se = st < NS ? st : NS - 1;
    char* slot = smem + (st & 3) * W4S_STAGE;
    if (g < 4) {
      glds16(ua + (size_t)g * a_p + (size_t)se * (W4S_BK * 2) + a_off, slot + (wave * 4 + g) * 1024);
    } else {
      const int p = g - 4;
      if constexpr (!B_KN) {
        glds16(ub + (size_t)p * a_p + (size_t)se * b_st + b_off, slot + 16384 + (wave * 4 + p) * 1024);
      } else {
        const int h = p >> 1, p2 = p & 1;
        glds16(ub + (size_t)p2 * b_q + (size_t)(256 * h) + (size_t)se * b_st + b_off,
               slot + 16384 + h * 8192 + (wave * 2 + p2) * 1024);
      }
    }
  };

  // ---- fragment read addresses inside a stage (one VGPR per k-step)
  const int q4 = (lane >> 2) & 3;   // (row>>2)&3 of this lane's fragment rows
  int a_ad[2], b_ad[4];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    a_ad[ks] = (wr * 128 + l32) * 64 + (((2 * ks + hi) ^ q4) * 16);
    if constexpr (!B_KN) b_ad[ks] = 16384 + (wc * 128 + l32) * 64 + (((2 * ks + hi) ^ q4) * 16);
  }
  if constexpr (B_KN) {
    const int i = lane & 15, gi = (lane >> 4) & 1;
    const int k = 8 * hi + (i >> 2);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      b_ad[j] = 16384 + wc * 8192 + k * 256 + (((2 * j + gi) ^ ((i >> 2) << 1)) * 32) + (i & 3) * 8;
  }
  auto read_a = [&](const char* slot, int ks, int i) -> half8_t {
    return *(const half8_t*)(slot + a_ad[ks] + i * 2048);
  };
  half4_t braw_lo, braw_hi;          // NN: raw halves of the transpose read just issued
  half4_t braw[2][8];                // NN: raw B fragments of fragment buffer 0 / 1
  auto read_b = [&](const char* slot, int ks, int j) -> half8_t {
    if constexpr (!B_KN) {
      return *(const half8_t*)(slot + b_ad[ks] + j * 2048);
    } else {
      // (asm transpose reads: hipcc would put s_waitcnt vmcnt(0) in front of the builtin form after every LDS-DMA)
      const uint32_t a = lds_addr32(slot + b_ad[j]) + (uint32_t)(ks * 4096);
      braw_lo = lds_tr16_asm<0>(a);
      braw_hi = lds_tr16_asm<1024>(a);
      return half8_t{};
    }
  };

  static_for<256>([&](auto r) { w4_acc_zero<decltype(r)::value>(); });

  // prologue: stages 0, 1, 2 and the first half of stage 3
#pragma unroll
  for (int st = 0; st < 3; ++st)
#pragma unroll
    for (int g = 0; g < 8; ++g) piece(g, st);
#pragma unroll
  for (int g = 0; g < 4; ++g) piece(g, 3);
  LC_VMCNT(20);
  pp_barrier();

  half8_t af[2][4], bf[2][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) af[0][i] = read_a(smem, 0, i);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    bf[0][j] = read_b(smem, 0, j);
    if constexpr (B_KN) {
      braw[0][2 * j] = braw_lo;
      braw[0][2 * j + 1] = braw_hi;
    }
  }

  // one k-step: 16 MFMAs from fragment buffer cb; chunks 0..3 carry the reads of the next k-step into
  // buffer cb^1, chunks 4..7 one DMA piece each (pieces g0..g0+3 of stage dst).
  auto step = [&](auto cbc, const char* rslot, int rks, int g0, int dst) {
    constexpr int cb = decltype(cbc)::value;
    if constexpr (B_KN) {   // the asm reads of this buffer were issued >= 8 MFMAs ago
      lds_tr16_wait8(braw[cb]);
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[cb][j] = cat4(braw[cb][2 * j], braw[cb][2 * j + 1]);
      // hipcc may build the operand tuples with v_mov copies right here: VALU write -> (asm) MFMA operand read
      asm volatile("s_nop 1" : "+v"(bf[cb][0]), "+v"(bf[cb][1]), "+v"(bf[cb][2]), "+v"(bf[cb][3]));
    }
    static_for<8>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      constexpr int i = c >> 1, j0 = 2 * (c & 1);
      w4_mfma<4 * i + j0>(bf[cb][j0], af[cb][i]);
      w4_mfma<4 * i + j0 + 1>(bf[cb][j0 + 1], af[cb][i]);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (c < 4) {
        af[cb ^ 1][c] = read_a(rslot, rks, c);
        bf[cb ^ 1][c] = read_b(rslot, rks, c);
        if constexpr (B_KN) {
          braw[cb ^ 1][2 * c] = braw_lo;
          braw[cb ^ 1][2 * c + 1] = braw_hi;
        }
      } else {
        piece(g0 + c - 4, dst);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  for (int st = 0; st < NS; ++st) {
    const char* cur = smem + (st & 3) * W4S_STAGE;
    const char* nxt = smem + ((st + 1) & 3) * W4S_STAGE;
    step(I0{}, cur, 1, 4, st + 3);
    asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
    pp_barrier();
    step(I1{}, nxt, 0, 0, st + 4);
  }
  LC_VMCNT(0);

  // ---- epilogue (as hgemm_w4_kernel)
  w4_mfma_drain();
  __syncthreads();
  char* stg = smem + wave * (32 * W4_EPI_STRIDE);
  half_t* cw = C + (size_t)(m0 + wr * 128) * N + n0 + wc * 128;
  static_for<4>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    static_for<16>([&](auto qc) {
      constexpr int j = decltype(qc)::value >> 2, rq = decltype(qc)::value & 3;
      half4_t h;
      h[0] = (half_t)w4_acc_read<16 * (4 * i + j) + 4 * rq + 0>();
      h[1] = (half_t)w4_acc_read<16 * (4 * i + j) + 4 * rq + 1>();
      h[2] = (half_t)w4_acc_read<16 * (4 * i + j) + 4 * rq + 2>();
      h[3] = (half_t)w4_acc_read<16 * (4 * i + j) + 4 * rq + 3>();
      *(half4_t*)(stg + l32 * W4_EPI_STRIDE + (j * 32 + 8 * rq + 4 * hi) * 2) = h;
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = it * 4 + (lane >> 4);
      const u32x4_t v = *(const u32x4_t*)(stg + row * W4_EPI_STRIDE + (lane & 15) * 16);
      *(u32x4_t*)(cw + (size_t)(i * 32 + row) * N + (lane & 15) * 8) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  });
}

}  // namespace lc
