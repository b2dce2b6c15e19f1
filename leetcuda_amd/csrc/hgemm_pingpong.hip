// hgemm_pingpong.hip — fp16 GEMM for gfx950, 256x256x64 tile, 8 wave64, phase-interleaved
// "ping-pong" schedule: the two waves that share a SIMD alternate between an MFMA cluster and a
// load section (LDS fragment reads + LDS-DMA issue), separated by raw s_barriers; global->LDS DMA
// stays in flight across barriers behind COUNTED s_waitcnt vmcnt(8) (never 0 inside the K loop).
//
// Same contract and LDS swizzles as hgemm_mfma256.hip (reference: kernels/hgemm/mma/basic/
// hgemm_mma_stage.cu:644-1052 NN, kernels/hgemm/mma/swizzle/hgemm_mma_stage_tn_swizzle_x4.cu:207 TN),
// v_mfma_f32_32x32x16_f16, swapped operands (accumulator = Cᵀ fragments).
//
// Wave (wr, wc) = (wave>>2, wave&3) owns C rows [128wr,+128) x cols [64wc,+64) of the tile as
// 4 (mt) x 2 (nt) 32x32 accumulators; the schedule itself (two phases of 16 MFMAs per K tile, hazards, DMA piece
// order) is described in front of hgemm_pingpong2_kernel below.  "half h" of A = rows with ((row>>6)&1)==h
// (16 KiB = 16 DMA pieces, 2 per wave); NN B: sub-image h = columns [128h, +128).
// This 8-wave kernel is the independently scheduled CROSS-CHECK of the default 4-wave kernel (hgemm_w4.hip); tests
// require the two to agree.  Its helpers (PPSrc/PPFrag, pp_barrier, LC_VMCNT) are shared with gemm_fp8.hip.
#pragma once
#include "hgemm_mfma256.hip"

namespace lc {

constexpr int HALF_BYTES = TILE_BYTES / 2;  // 16 KiB

#define LC_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

template <bool B_KN>
struct PPSrc {
  const half_t* a[2][2];  // [half][piece]
  const half_t* b[2][2];
  int a_lds[2][2];        // LDS byte offset of the piece inside a slot
  int b_lds[2][2];
};

template <bool B_KN>
LC_DEVINL void pp_src_init(PPSrc<B_KN>& s, const half_t* A, const half_t* B, int m0, int n0, int N,
                           int K, int wave, int lane) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = wave * 2 + i;  // 0..15: this wave's piece inside the half
      {  // A: 8-row blocks; half h of group g = blocks 16g + 8h + j
        const int blk = 16 * (q >> 3) + 8 * h + (q & 7);
        const int row = blk * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        s.a[h][i] = A + (size_t)(m0 + row) * K + c * 8;
        s.a_lds[h][i] = blk * 1024;
      }
      if constexpr (!B_KN) {  // B stored [N][K]: half h of wave-column wc = blocks 8wc + 4h + j
        const int blk = 8 * (q >> 2) + 4 * h + (q & 3);
        const int row = blk * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        s.b[h][i] = B + (size_t)(n0 + row) * K + c * 8;
        s.b_lds[h][i] = TILE_BYTES + blk * 1024;
      } else {  // B stored [K][N]: sub-image h = [64 k][128 n'] (256 B rows), piece = 4 k rows
        const int k = q * 4 + (lane >> 4);
        const int pp = lane & 15;
        const int pair = (pp >> 1) ^ ((k & 3) << 1);
        const int nc = pair * 2 + (pp & 1);  // 16-B chunk index inside the sub-image row
        // sub-image h = the 128 CONTIGUOUS columns [128h, +128) -> every DMA lane group fetches whole 128-B lines
        // (measured: 64-B source segments double the TA time of a DMA piece)
        const int n = 128 * h + nc * 8;
        s.b[h][i] = B + (size_t)k * N + n0 + n;
        s.b_lds[h][i] = TILE_BYTES + h * HALF_BYTES + q * 1024;
      }
    }
  }
}

template <bool B_KN>
struct PPFrag {
  int a0;  // A: row (wr*128 + l32), chunk (hi ^ swz); + mt*4096, ^ ks*32
  int b0;  // TN B: row (wc*64 + l32); + nh*4096, ^ ks*32 ; NN B: tr base of n-half 0; + ks*4096 (+1024)
  int b1;  // NN B: tr base of n-half 1
};

template <bool B_KN>
LC_DEVINL void pp_frag_init(PPFrag<B_KN>& f, int wr, int wc, int lane) {
  const int l32 = lane & 31, hi = lane >> 5;
  const int swz = (lane >> 1) & 7;
  f.a0 = (wr * 128 + l32) * 128 + ((hi ^ swz) * 16);
  if constexpr (!B_KN) {
    f.b0 = TILE_BYTES + (wc * 64 + l32) * 128 + ((hi ^ swz) * 16);
    f.b1 = 0;
  } else {
    const int i = lane & 15, gi = (lane >> 4) & 1;
    const int k = 8 * hi + (i >> 2);
    {   // columns 64wc + 32nh + 16gi.. live in sub-image wc>>1 at 32-B pair 4(wc&1) + 2nh + gi
      const int base = TILE_BYTES + (wc >> 1) * HALF_BYTES + k * 256 + (i & 3) * 8;
      f.b0 = base + (((4 * (wc & 1) + gi) ^ ((i >> 2) << 1)) * 32);
      f.b1 = base + (((4 * (wc & 1) + 2 + gi) ^ ((i >> 2) << 1)) * 32);
    }
  }
}

template <bool B_KN>
LC_DEVINL void pp_read_a(const char* slot, const PPFrag<B_KN>& f, int mh, half8_t (&af)[2][4]) {
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      af[m][ks] = *(const half8_t*)(slot + ((f.a0 ^ (ks * 32)) + (mh * 2 + m) * 4096));
}

template <bool B_KN>
LC_DEVINL void pp_read_b(const char* slot, const PPFrag<B_KN>& f, int nh, half8_t (&bf)[4]) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    if constexpr (!B_KN) {
      bf[ks] = *(const half8_t*)(slot + ((f.b0 ^ (ks * 32)) + nh * 4096));
    } else {
      const char* p = slot + (nh ? f.b1 : f.b0) + ks * 4096;
      bf[ks] = cat4(lds_tr16(p), lds_tr16(p + 1024));
    }
  }
}

// NN B fragments with ASM transpose reads (lc_common.h lds_tr16_asm: the builtin form makes hipcc drain the whole
// LDS-DMA prefetch — s_waitcnt vmcnt(0) — in front of the first transpose read of every K tile).  Two steps: issue
// the 8 reads of n-half nh into raw[], later pp_b_nn_finish() waits for them and forms the MFMA operands.
template <int NH>
LC_DEVINL void pp_read_b_nn_issue(const char* slot, const PPFrag<true>& f, half4_t (&raw)[8]) {
  const uint32_t a = lds_addr32(slot + (NH ? f.b1 : f.b0));
  raw[0] = lds_tr16_asm<0>(a);
  raw[1] = lds_tr16_asm<1024>(a);
  raw[2] = lds_tr16_asm<4096>(a);
  raw[3] = lds_tr16_asm<4096 + 1024>(a);
  raw[4] = lds_tr16_asm<8192>(a);
  raw[5] = lds_tr16_asm<8192 + 1024>(a);
  raw[6] = lds_tr16_asm<12288>(a);
  raw[7] = lds_tr16_asm<12288 + 1024>(a);
}
LC_DEVINL void pp_b_nn_finish(half4_t (&raw)[8], half8_t (&bf)[4]) {
  lds_tr16_wait8(raw);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) bf[ks] = cat4(raw[2 * ks], raw[2 * ks + 1]);
}

LC_DEVINL void pp_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

// Epilogue for the 32x32 swapped-accumulator layout: lane holds C[m = l32][n = 8*(r>>2)+4*hi+(r&3)].
LC_DEVINL void pp_epilogue(char* smem, f32x16_t (&acc)[4][2], half_t* C, int N, int m0, int n0,
                           int wave, int wr, int wc, int lane) {
  char* stg = smem + wave * (64 * EPI_STRIDE);
  const int l32 = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          half4_t h;
#pragma unroll
          for (int j = 0; j < 4; ++j) h[j] = (half_t)acc[pass * 2 + m][nt][4 * rq + j];
          *(half4_t*)(stg + (m * 32 + l32) * EPI_STRIDE + (nt * 32 + 8 * rq + 4 * hi) * 2) = h;
        }
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

// ------------------------------------------------------------------------------------------------
// Two-phase ping-pong (LC_HGEMM_MFMA256P2): 16 MFMAs per phase, 4 barriers per K tile, LDS-DMA issued
// from INSIDE the MFMA clusters (between MFMAs, where the wave has spare issue slots) instead of from the
// load sections.
//   phase A(kt): reads A0 (8 b128), B0 (4), B1 (4)            -> quadrants (0,0), (0,1)
//                MFMA cluster carries the 2 DMA pieces of A1(kt+1)
//   phase B(kt): reads A1 (8)                                 -> quadrants (1,0), (1,1)
//                MFMA cluster carries the 6 DMA pieces of B0, B1, A0 of tile kt+2
// Waits (issue order ... A0B(kt) | A1(kt) | A0B(kt+1) | A1(kt+1) ...):
//   end of load A(kt): vmcnt(6) -> A1(kt) landed   (6 younger: B0,B1,A0 of kt+1)
//   end of load B(kt): vmcnt(2) -> B0,B1,A0(kt+1) landed (2 younger: A1(kt+1))
// RAW: every half is waited for one phase before it is read (+ a barrier). WAR (slot = barrier interval;
// group 0 loads phase j in slot 2j and computes in 2j+1, group 1 one slot later): a half last read in
// load phase j is dead from slot 2j+3; A0/B of tile kt (j = 2kt) are re-staged in MFMA(2kt+1) = slots
// 4kt+3 / 4kt+4, A1 (j = 2kt+1) in MFMA(2kt+2) = slots 4kt+5 / 4kt+6.
template <bool B_KN, int NG, typename IssueFn>
LC_DEVINL void pp2_mfma(f32x16_t (&acc)[4][2], int mh, const half8_t (&af)[2][4], const half8_t (&b0f)[4],
                        const half8_t (&b1f)[4], IssueFn issue) {
  __builtin_amdgcn_s_setprio(1);
  // k-step outermost: the 4 accumulators of the phase rotate, so an accumulator is reused only every 4th
  // MFMA (a dependent v_mfma_f32_32x32x16 issued 2 slots after its producer still stalls ~10 cycles:
  // measured 676 instead of 512 cycles per 16-MFMA cluster with the 2-accumulator order)
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const int ks = g >> 1, nh = g & 1;
    const half8_t bf = nh ? b1f[ks] : b0f[ks];
    acc[mh * 2 + 0][nh] = mfma32(bf, af[0][ks], acc[mh * 2 + 0][nh]);
    acc[mh * 2 + 1][nh] = mfma32(bf, af[1][ks], acc[mh * 2 + 1][nh]);
    if (g < NG) issue(g);
    __builtin_amdgcn_sched_barrier(0);
  }
  __builtin_amdgcn_s_setprio(0);
}

// (measured and dropped in round 1: issuing the DMA from the load sections instead — 2 barrier intervals of flight are
// not enough, 1187 vs 1322 TFLOP/s — and the 4-phase / 8-barrier schedule, 1283)
template <bool B_KN, bool STAMPS = false>
__global__ __launch_bounds__(512, 2) void hgemm_pingpong2_kernel(const half_t* __restrict__ A,
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

  PPSrc<B_KN> src;
  pp_src_init<B_KN>(src, A, B, m0, n0, N, K, wave, lane);
  PPFrag<B_KN> fr;
  pp_frag_init<B_KN>(fr, wr, wc, lane);

  f32x16_t acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int KT = K / BK;
  const size_t bstep = B_KN ? (size_t)BK * N : (size_t)BK;
  // one DMA piece: operand (0 = A, 1 = B), half h, piece i of this wave, K tile t (clamped) -> slot t&1
  auto piece = [&](int is_b, int h, int i, int t) {
    const int te = t < KT ? t : KT - 1;
    char* slot = smem + (t & 1) * SLOT_BYTES;
    if (is_b)
      glds16(src.b[h][i] + (size_t)te * bstep, slot + src.b_lds[h][i]);
    else
      glds16(src.a[h][i] + (size_t)te * BK, slot + src.a_lds[h][i]);
  };
  // g = 0..5 -> B0[0] B0[1] B1[0] B1[1] A0[0] A0[1]
  auto issue_ab0 = [&](int g, int t) { piece(g < 4, g < 4 ? (g >> 1) : 0, g & 1, t); };

  // prologue: A0B(0) | A1(0) [| A0B(1)]
#pragma unroll
  for (int g = 0; g < 6; ++g) issue_ab0(g, 0);
  piece(0, 1, 0, 0);
  piece(0, 1, 1, 0);
#pragma unroll
  for (int g = 0; g < 6; ++g) issue_ab0(g, 1);
  LC_VMCNT(8);
  pp_barrier();
  if (wr == 1) pp_barrier();

  // STAMPS (diagnosis only, clobbers the first bytes of A): cycle stamps of waves 0 and 4 of workgroup 0 at
  // the phase boundaries of K tiles 32..35
  unsigned long long* stamp = reinterpret_cast<unsigned long long*>(const_cast<half_t*>(A));
  const bool stamping = STAMPS && blockIdx.x == 0 && (wave == 0 || wave == 4) && lane == 0;
  auto STAMP = [&](int kt, int k) {
    if constexpr (STAMPS) {
      if (kt >= 32 && kt < 36) {
        const unsigned long long c = __builtin_readcyclecounter();
        if (stamping) stamp[((wave >> 2) * 4 + (kt - 32)) * 8 + k] = c;
      }
    }
  };
  half8_t af[2][4], b0f[4], b1f[4];
  for (int kt = 0; kt < KT; ++kt) {
    const char* cur = smem + (kt & 1) * SLOT_BYTES;
    STAMP(kt, 0);
    // ---- phase A
    half4_t braw0[8], braw1[8];
    if constexpr (B_KN) {
      pp_read_b_nn_issue<0>(cur, fr, braw0);
      pp_read_a<B_KN>(cur, fr, 0, af);
      pp_read_b_nn_issue<1>(cur, fr, braw1);
      pp_b_nn_finish(braw0, b0f);
      pp_b_nn_finish(braw1, b1f);
    } else {
      pp_read_b<B_KN>(cur, fr, 0, b0f);
      pp_read_a<B_KN>(cur, fr, 0, af);
      pp_read_b<B_KN>(cur, fr, 1, b1f);
    }
    if constexpr (STAMPS) {   // split "fragment reads returned" from "DMA landed"
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      STAMP(kt, 5);
    }
    LC_VMCNT(6);
    STAMP(kt, 1);
    pp_barrier();
    STAMP(kt, 2);
    pp2_mfma<B_KN, 2>(acc, 0, af, b0f, b1f, [&](int g) { piece(0, 1, g, kt + 1); });
    STAMP(kt, 3);
    pp_barrier();
    STAMP(kt, 4);
    // ---- phase B
    pp_read_a<B_KN>(cur, fr, 1, af);
    LC_VMCNT(2);
    pp_barrier();
    STAMP(kt, 6);
    pp2_mfma<B_KN, 6>(acc, 1, af, b0f, b1f, [&](int g) { issue_ab0(g, kt + 2); });
    STAMP(kt, 7);
    pp_barrier();
  }
  if (wr == 0) pp_barrier();
  LC_VMCNT(0);
  pp_epilogue(smem, acc, C, N, m0, n0, wave, wr, wc, lane);
}


}  // namespace lc
