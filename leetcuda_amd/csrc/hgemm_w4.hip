// hgemm_w4.hip — fp16 GEMM for gfx950, 256x256x64 tile, FOUR wave64 (2 x 2), wave tile 128 x 128: the kernel
// 32x32x16-MFMA baseline of the 4-wave family (LC_HGEMM_MFMA256W4B / W4C; LC_HGEMM_AUTO until round 2's 16x16x32 kernels
// hgemm_w4x.hip / hgemm_w4y.hip, which keep this file's ring, DMA and swizzles).
//
// Same contract, LDS images and swizzles as hgemm_pingpong.hip (reference: kernels/hgemm/mma/basic/
// hgemm_mma_stage.cu:644-1052 NN, kernels/hgemm/mma/swizzle/hgemm_mma_stage_tn_swizzle_x4.cu:207 TN).
// Why a second 256-tile design: the 8-wave kernels run the chip into its POWER limit (80 % MFMA-busy at
// ~1.6 GHz), so throughput is bought with energy per FLOP, not with more busy cycles.  A 128x128 wave tile
// reads (128+128) x 16 B per 16 MFMAs = 0.5 KiB of LDS per MFMA instead of 0.75 KiB (128x64), and one wave
// per SIMD needs no s_setprio ping-pong and only ONE barrier per K tile:
//   * one wave per SIMD (256 threads, 512-entry register file): 4x4 v_mfma_f32_32x32x16_f16
//     accumulators = 256 registers (AGPR half, addressed literally), fragments double-buffered in 64 VGPRs;
//   * the K tile is 4 k-steps of 16 MFMAs; the fragment reads of step s+1 (8 ds_read_b128 / 16 tr reads)
//     are issued in the first half of step s, behind MFMAs of the same wave (software pipeline inside one
//     instruction stream, chunked with sched_barrier);
//   * LDS: A ring of 2 K tiles + B ring of 3 K tiles (5 x 32 KiB = 160 KiB, all of a CU's LDS).  The B pieces of
//     tile t+2 go out during steps 1-2 of tile t (their slot has been dead since the end of tile t-1), one per
//     4 MFMAs, and are in flight for >= 5 k-steps; the A pieces of tile t+2 follow behind the barrier — no DMA
//     piece is issued less than 1500 MFMA cycles before it is needed, and the wait in front of the barrier is a
//     counted vmcnt(8) (the 8 B pieces of tile t+2 stay in flight across it):
//       issue order  ... | B(t+1): steps 1-2 of t-1 | A(t+1) | B(t+2): steps 1-2 of t | A(t+2) ...
//     All 8 A pieces ride in step 3 (the step behind the barrier); spreading them over steps 3 / 0 (round 2's W4D)
//     or staggering the waves' issue (W4E) measured neutral / slower and is retired (git history).
//   * BUF: the DMA pieces are buffer_load ... lds (descriptor from a readfirstlane'd pointer + scalar offset + one
//     32-bit lane offset) instead of global_load_lds with a 64-bit address per lane.
//   * NN B image: two sub-images of 128 CONTIGUOUS columns ([64 k][256 B] each, 32-B pairs XOR-ed by (k&3)<<1,
//     read with ds_read_b64_tr_b16): every DMA lane group fetches whole 128-B lines.
// Measured ladder (8192^3 TN, one box, round 1): 2-slot ring 1315 -> 3-slot B ring 1316 -> + buffer DMA 1341 TFLOP/s;
// ablation (LC_DIAG, tools/w4_ablate.py): MFMA issue only 0.665 ms = 1650 TFLOP/s — the ceiling of this chip on
// randn data at its power-limited clock; + fragment reads 0.71; + barrier 0.72; + DMA 0.82.
// Retired round-1 siblings (git history): the 2-slot ring (w4) and the 4-stage/32-k ring (w4s, 64-B DMA rows cost
// twice the TA time).
#pragma once
#include "hgemm_pingpong.hip"

namespace lc {

constexpr int W4B_LDS = 5 * TILE_BYTES;   // A ring 2 x 32 KiB + B ring 3 x 32 KiB
constexpr int W4_EPI_STRIDE = 272;        // bytes per staged C row (128 halves + 16 B pad)

// The 4x4 accumulator tile lives in a[0:255], addressed LITERALLY: hipcc's allocator cannot hold 256 live
// accumulator registers (as values — builtin or "+a" operands — it emits ~1500 v_accvgpr copies and 250 scratch
// accesses per K tile).  Every statement names all 256 AGPRs as clobbered, so the compiler never parks a value of
// its own there; leetcuda_amd/build.py audits the emitted ISA after every build (no v_accvgpr_* outside
// ASMSTART/ASMEND, no scratch, no compiler access to an asm transpose-read destination before its wait).
// Hazards handled here, not by hipcc: accumulate chains (D -> C of the next MFMA on the same registers) need no
// wait states; A/B fragments come from compiler-visible ds_reads (hipcc waits for them); the first
// v_accvgpr_read of the epilogue is fenced by w4_mfma_drain() (>= 12 states after an 8-pass MFMA).
template <int IDX0, int IDX1>
LC_DEVINL void w4_mfma2(half8_t a0, half8_t a1, half8_t b) {   // two MFMAs sharing operand b, one statement (no pad between)
  asm volatile("v_mfma_f32_32x32x16_f16 a[%3:%4], %0, %2, a[%3:%4]\n\tv_mfma_f32_32x32x16_f16 a[%5:%6], %1, %2, a[%5:%6]"
               :: "v"(a0), "v"(a1), "v"(b), "n"(IDX0 * 16), "n"(IDX0 * 16 + 15), "n"(IDX1 * 16), "n"(IDX1 * 16 + 15)
               : LC_AGPR_ALL);
}
template <int R>
LC_DEVINL void w4_acc_zero() { asm volatile("v_accvgpr_write_b32 a[%0], 0" :: "n"(R) : LC_AGPR_ALL); }
// LC_AGPR_ALL on the read as well: the statement then conflicts with any compiler value parked in an AGPR, so hipcc
// can never keep one of its own there between two statements (the read used to carry no clobber).
template <int R>
LC_DEVINL float w4_acc_read() {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(x) : "n"(R) : LC_AGPR_ALL);
  return x;
}
template <bool SCALE>
LC_DEVINL float w4_scaled(float x, float alpha) {
  if constexpr (SCALE) return x * alpha;
  else return x;
}
LC_DEVINL void w4_mfma_drain() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }

// ---- DMA sources of one wave.  A (and TN B): the wave stages the 8-row blocks blk = 8*wave + p, p = 0..7; lane ->
// row (lane>>3) of the block, chunk slot (lane&7) holding logical chunk slot ^ ((row>>1)&7), and
// (row>>1)&7 = ((lane>>4)&3) | ((p&1)<<2): 2 offset VGPRs per operand instead of 16 pointers.
// NN B: sub-image h = columns [128h, +128) as [64 k][256 B], piece q = 4 k-rows; the wave stages q = 4*wave + p2 of
// both sub-images; lane -> k row (lane>>4), 16-B slot pp = lane&15 holding chunk nc (32-B pairs XOR-ed by (k&3)<<1).
template <bool B_KN>
struct W4Src {
  const char* ua;
  const char* ub;
  unsigned a_off[2], b_off[2];
  size_t a_blk, a_kt, b_kt, b_q;
};
template <bool B_KN>
LC_DEVINL void w4_src_init(W4Src<B_KN>& s, const half_t* A, const half_t* B, int m0, int n0, int N, int K, int wave,
                           int lane) {
#pragma unroll
  for (int par = 0; par < 2; ++par)
    s.a_off[par] = (unsigned)(lane >> 3) * (unsigned)K * 2u +
                   (unsigned)(((lane & 7) ^ (((lane >> 4) & 3) | (par << 2))) * 16);
  s.ua = (const char*)(A + (size_t)(m0 + wave * 64) * K);
  if constexpr (!B_KN) {
    s.ub = (const char*)(B + (size_t)(n0 + wave * 64) * K);
    s.b_off[0] = s.a_off[0];
    s.b_off[1] = s.a_off[1];
  } else {
    const int pp = lane & 15;
    const int pair = (pp >> 1) ^ (((lane >> 4) & 3) << 1);
    const int nc = pair * 2 + (pp & 1);
    s.b_off[0] = (unsigned)(lane >> 4) * (unsigned)N * 2u + (unsigned)(nc * 16);
    s.b_off[1] = s.b_off[0];
    s.ub = (const char*)(B + (size_t)(wave * 16) * N + n0);
  }
  s.a_blk = (size_t)8 * K * 2;                        // bytes between consecutive 8-row blocks
  s.a_kt = (size_t)BK * 2;                            // bytes per K tile along a row
  s.b_kt = B_KN ? (size_t)BK * N * 2 : (size_t)BK * 2;
  s.b_q = (size_t)4 * N * 2;                          // NN: bytes between consecutive pieces
}

// ---- fragment read addresses (one VGPR per k-step: no address arithmetic in the loop), relative to the A / B slot
template <bool B_KN>
struct W4Frag {
  int a_ad[4], b_ad[4];
};
template <bool B_KN>
LC_DEVINL void w4_frag_init(W4Frag<B_KN>& f, int wr, int wc, int lane) {
  const int l32 = lane & 31, hi = lane >> 5;
  const int swz = (lane >> 1) & 7;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    f.a_ad[ks] = (wr * 128 + l32) * 128 + (((2 * ks + hi) ^ swz) * 16);
    if constexpr (!B_KN) f.b_ad[ks] = (wc * 128 + l32) * 128 + (((2 * ks + hi) ^ swz) * 16);
  }
  if constexpr (B_KN) {
    const int i = lane & 15, gi = (lane >> 4) & 1;
    const int k = 8 * hi + (i >> 2);
#pragma unroll
    for (int j = 0; j < 4; ++j)   // n-block j = columns 128wc + 32j + 16gi.. = pair 2j + gi of sub-image wc
      f.b_ad[j] = wc * HALF_BYTES + k * 256 + (((2 * j + gi) ^ ((i >> 2) << 1)) * 32) + (i & 3) * 8;
  }
}

// ---- epilogue: lane holds C[m = 32i + l32][n = 32j + 8(r>>2) + 4hi + (r&3)] in a[16(4i+j) + r]; each wave stages one
// 32-row block (32 x 128 halves) at a time in its private LDS area and writes whole 256-B row segments (16-B stores).
// SCALE: multiply by alpha in fp32 first (the fp8 GEMM of gemm_fp8.hip shares this epilogue).
template <bool SCALE = false>
LC_DEVINL void w4_epilogue(char* smem, half_t* C, int N, int m0, int n0, int wave, int wr, int wc, int lane,
                           float alpha = 1.0f) {
  const int l32 = lane & 31, hi = lane >> 5;
  w4_mfma_drain();
  __syncthreads();
  char* stg = smem + wave * (32 * W4_EPI_STRIDE);
  half_t* cw = C + (size_t)(m0 + wr * 128) * N + n0 + wc * 128;
  static_for<4>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    static_for<16>([&](auto qc) {
      constexpr int j = decltype(qc)::value >> 2, rq = decltype(qc)::value & 3;
      half4_t h;
      h[0] = (half_t)w4_scaled<SCALE>(w4_acc_read<16 * (4 * i + j) + 4 * rq + 0>(), alpha);
      h[1] = (half_t)w4_scaled<SCALE>(w4_acc_read<16 * (4 * i + j) + 4 * rq + 1>(), alpha);
      h[2] = (half_t)w4_scaled<SCALE>(w4_acc_read<16 * (4 * i + j) + 4 * rq + 2>(), alpha);
      h[3] = (half_t)w4_scaled<SCALE>(w4_acc_read<16 * (4 * i + j) + 4 * rq + 3>(), alpha);
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

// DG (LC_DIAG builds only; results are WRONG when an ablation bit is set):
//   1 = cycle stamps (clobbers the first bytes of A): s_memtime of wave 0 of workgroup 0 at the step boundaries of K
//       tiles 32..35 (lc_tune_set "hgemm_stamps", tools/hgemm_w4c_stamps.py)
//   2 = no DMA after the prologue, 4 = no per-tile wait + barrier, 8 = no fragment reads in the loop (2|4|8 = MFMA
//       issue only: the ceiling of the matrix pipe at the sustained clock; lc_tune_set "w4_abl", tools/w4_ablate.py)
template <bool B_KN, bool BUF, int DG = 0>
__global__ __launch_bounds__(256) void hgemm_w4b_kernel(const half_t* __restrict__ A,
                                                       const half_t* __restrict__ B,
                                                       half_t* __restrict__ C, int M, int N, int K,
                                                       int tiles_m, int tiles_n, int panel_w) {
  constexpr bool STAMPS = (DG & 1) != 0;
  constexpr bool NO_DMA = (DG & 2) != 0, NO_BAR = (DG & 4) != 0, NO_READ = (DG & 8) != 0;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = wave_id();
  const int wr = wave >> 1, wc = wave & 1;

  const TileCoord tc = block_tile(blockIdx.x, gridDim.x, tiles_m, tiles_n, panel_w);
  const int m0 = tc.tm * BM, n0 = tc.tn * BN;

  W4Src<B_KN> src;
  w4_src_init<B_KN>(src, A, B, m0, n0, N, K, wave, lane);
  // piece g = 0..15 of K tile t (clamped: past the end the last tile is staged again into a dead slot, which keeps
  // the vmcnt counts exact) -> `slot` = the A slot (g < 8) or the B slot (g >= 8) of that tile
  const int KT = K / BK;
  const buf_rsrc_t ra = make_rsrc(src.ua), rb = make_rsrc(src.ub);
  auto dma = [&](buf_rsrc_t r, const char* base, size_t uoff, unsigned loff, char* dst) {
    if constexpr (BUF) blds16(r, loff, (unsigned)uoff, dst);
    else glds16(base + uoff + loff, dst);
  };
  auto piece = [&](int g, int t, char* slot) {
    const int te = t < KT ? t : KT - 1;
    if (g < 8) {
      dma(ra, src.ua, (size_t)g * src.a_blk + (size_t)te * src.a_kt, src.a_off[g & 1], slot + (wave * 8 + g) * 1024);
    } else {
      const int p = g - 8;
      if constexpr (!B_KN) {
        dma(rb, src.ub, (size_t)p * src.a_blk + (size_t)te * src.b_kt, src.b_off[p & 1], slot + (wave * 8 + p) * 1024);
      } else {
        const int h = p >> 2, p2 = p & 3;
        dma(rb, src.ub, (size_t)p2 * src.b_q + (size_t)(256 * h) + (size_t)te * src.b_kt, src.b_off[0],
            slot + h * HALF_BYTES + (wave * 4 + p2) * 1024);
      }
    }
  };
  auto a_slot = [&](int t) -> char* { return smem + (t & 1) * TILE_BYTES; };
  auto b_slot = [&](int bi) -> char* { return smem + (2 + bi) * TILE_BYTES; };   // bi = t % 3, tracked by the caller

  W4Frag<B_KN> fr;
  w4_frag_init<B_KN>(fr, wr, wc, lane);
  auto read_a = [&](const char* slot, int ks, int i) -> half8_t {
    return *(const half8_t*)(slot + fr.a_ad[ks] + i * 4096);
  };
  half4_t braw_lo, braw_hi;          // NN: raw halves of the transpose read just issued
  half4_t braw[2][8];                // NN: raw B fragments of fragment buffer 0 / 1
  auto read_b = [&](const char* slot, int ks, int j) -> half8_t {
    if constexpr (!B_KN) {
      return *(const half8_t*)(slot + fr.b_ad[ks] + j * 4096);
    } else {
      // (asm transpose reads: hipcc would put s_waitcnt vmcnt(0) in front of the builtin form after every LDS-DMA)
      const uint32_t a = lds_addr32(slot + fr.b_ad[j]) + (uint32_t)(ks * 4096);
      braw_lo = lds_tr16_asm<0>(a);
      braw_hi = lds_tr16_asm<1024>(a);
      return half8_t{};
    }
  };

  static_for<256>([&](auto r) { w4_acc_zero<decltype(r)::value>(); });   // accumulator (i, j) = a[16(4i+j) ..]

  // prologue: B(0) A(0) B(1) A(1); tile 0 landed, step-0 fragments of tile 0
  // in registers
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
  auto step = [&](auto cbc, const char* rsa, const char* rsb, int rks, auto npc, int g0, int t2, char* wslot) {
    constexpr int cb = decltype(cbc)::value;
    constexpr int NP = decltype(npc)::value;
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
      // MFMAs FIRST: hipcc guards the first asm consumer of the previous step's fragments with lgkmcnt(0); ahead of
      // this step's reads that wait is free, behind them it would expose a full LDS round trip per step.
      w4_mfma2<4 * i + j0, 4 * i + j0 + 1>(bf[cb][j0], bf[cb][j0 + 1], af[cb][i]);
      __builtin_amdgcn_sched_barrier(0);
      // (one read per chunk over all 8 chunks instead of two in chunks 0..3 measures the same: A/B run r04c)
      if constexpr (c < 4 && !NO_READ) {
        af[cb ^ 1][c] = read_a(rsa, rks, c);
        bf[cb ^ 1][c] = read_b(rsb, rks, c);
        if constexpr (B_KN) {
          braw[cb ^ 1][2 * c] = braw_lo;
          braw[cb ^ 1][2 * c + 1] = braw_hi;
        }
      }
      if constexpr (!NO_DMA) {
        if constexpr (NP == 8) piece(g0 + c, t2, wslot);
        if constexpr (NP == 4 && (c & 1) == 1) piece(g0 + (c >> 1), t2, wslot);
      }
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
    if constexpr (!NO_BAR) {
      asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      STAMP(kt, 4);
      pp_barrier();
    }
    STAMP(kt, 5);
    step(I1{}, a_slot(kt + 1), b_slot(b1), 0, P8{}, 0, kt + 2, a_slot(kt));
    STAMP(kt, 6);
    const int t = b0;
    b0 = b1;
    b1 = b2;
    b2 = t;
  }
  LC_VMCNT(0);
  if constexpr (B_KN) {
    // the last step issued (dead) asm transpose reads of a tile past the end: retire them before hipcc may reuse their
    // destination registers in the epilogue (found by leetcuda_amd/isa_audit.py rule R3)
    lds_tr16_wait8(braw[0]);
    lds_tr16_wait8(braw[1]);
  }
  w4_epilogue(smem, C, N, m0, n0, wave, wr, wc, lane);
}

}  // namespace lc
