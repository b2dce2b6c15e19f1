// hgemm_w4x.hip — fp16 GEMM for gfx950, 256x256x64 tile, four wave64 with 128x128 wave tiles: the ring / DMA / barrier
// skeleton of hgemm_w4b_kernel<B_KN, BUF = true> (hgemm_w4.hip) with v_mfma_f32_16x16x32_f16 instead of
// v_mfma_f32_32x32x16_f16 (LC_HGEMM_MFMA256W4X).
//
// Why (round 2, profiles/r2_power_*.log, tools/cpp/mfma_power.cpp + tools/power_watch.sh): at 8192^3 on random data the
// 32x32x16 kernel sits at the board's 1400 W cap at 1.57-1.67 GHz — its throughput is set by JOULES PER FLOP, not by
// issue slots (the same binary on zero-filled operands runs 2.39 GHz / 2035 TFLOP/s at 1180-1260 W).  Measured with an
// MFMA-only stream on N(0,1) operands at the cap: 32x32x16 sustains 1625 TFLOP/s (1.65-1.71 GHz), 16x16x32 1854
// (1.90-1.93 GHz): +14 % per joule — the 16x16 form moves half the accumulator registers per FLOP (4 in / 4 out per
// 16 KFLOP against 16 / 16 per 32 KFLOP), and the accumulator file is the widest operand of an MFMA.  With the GEMM's
// LDS-read and L2 -> LDS traffic added the ratio stays (1438 vs 1255).  hipBLASLt's kernel for this shape
// (Custom_Cijk_Alik_Bljk_HHS_BH_MT256x256x64_MI16x16x1) makes the same choice.
//
// Same contract, LDS images, source-side swizzles and DMA schedule as hgemm_w4b_kernel (reference: kernels/hgemm/mma/
// basic/hgemm_mma_stage.cu:644-1052 NN, kernels/hgemm/mma/swizzle/hgemm_mma_stage_tn_swizzle_x4.cu:207 TN).  What
// changes:
//   * accumulators: 8 x 8 blocks of 16 x 16, block (i, j) = a[4(8i+j) .. +3] (256 AGPRs, addressed literally);
//     MFMA(SrcA = B fragment j, SrcB = A fragment i): lane holds C[m = 16i + (l & 15)][n = 16j + 4(l >> 4) + r];
//   * a K tile (64) is 2 k-steps of 32; a k-step needs 8 A + 8 B fragments (ds_read_b128 each: lane -> row l & 15,
//     16-B chunk 4ks + (l >> 4) — the [rows][128 B] image with chunk ^= (row >> 1) & 7 is conflict-free for these lane
//     groups too: tests/test_layouts.py) and issues 64 MFMAs;
//   * the loop keeps FOUR steps of 512 MFMA cycles per K tile: step s = (k-step s >> 1, A-row half s & 1), 32 MFMAs.
//     A fragments are double-buffered per step (2 x 4), B fragments per k-step (2 x 8); a step reads the 4 A fragments
//     of the next step, the odd steps also the 8 B fragments of the next k-step (step 3: of the NEXT tile — behind the
//     barrier, like hgemm_w4b's step 3).  The DMA pieces ride where they did: B(t+2) in steps 1-2, A(t+2) in step 3.
#pragma once
#include "hgemm_w4.hip"

namespace lc {

// two 16x16x32 MFMAs sharing SrcB (= the A fragment), one statement (hipcc pads nothing between them)
template <int BLK0, int BLK1>
LC_DEVINL void w4x_mfma2(half8_t b0, half8_t b1, half8_t a) {
  asm volatile("v_mfma_f32_16x16x32_f16 a[%3:%4], %0, %2, a[%3:%4]\n\tv_mfma_f32_16x16x32_f16 a[%5:%6], %1, %2, a[%5:%6]"
               :: "v"(b0), "v"(b1), "v"(a), "n"(BLK0 * 4), "n"(BLK0 * 4 + 3), "n"(BLK1 * 4), "n"(BLK1 * 4 + 3)
               : LC_AGPR_ALL);
}

// fragment read addresses relative to the A / B slot: one VGPR per k-step; fragment i (16 rows) at + i * 2048
struct W4xFrag {
  int a_ad[2], b_ad[2];
};
LC_DEVINL void w4x_frag_init(W4xFrag& f, int wr, int wc, int lane) {
  const int r16 = lane & 15, kg = lane >> 4;
  const int swz = (lane >> 1) & 7;   // (row >> 1) & 7 of row = 16 i + r16
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    f.a_ad[ks] = (wr * 128 + r16) * 128 + (((4 * ks + kg) ^ swz) * 16);
    f.b_ad[ks] = (wc * 128 + r16) * 128 + (((4 * ks + kg) ^ swz) * 16);
  }
}

// epilogue: 32-row blocks (A fragments 2q, 2q+1) staged in the wave's private LDS area, whole 256-B row segments out
LC_DEVINL void w4x_epilogue(char* smem, half_t* C, int N, int m0, int n0, int wave, int wr, int wc, int lane) {
  const int r16 = lane & 15, kg = lane >> 4;
  w4_mfma_drain();
  __syncthreads();
  char* stg = smem + wave * (32 * W4_EPI_STRIDE);
  half_t* cw = C + (size_t)(m0 + wr * 128) * N + n0 + wc * 128;
  static_for<4>([&](auto qc) {
    constexpr int q = decltype(qc)::value;
    static_for<16>([&](auto ec) {
      constexpr int ih = decltype(ec)::value >> 3, j = decltype(ec)::value & 7;
      constexpr int base = 4 * (8 * (2 * q + ih) + j);
      half4_t h;
      h[0] = (half_t)w4_acc_read<base + 0>();
      h[1] = (half_t)w4_acc_read<base + 1>();
      h[2] = (half_t)w4_acc_read<base + 2>();
      h[3] = (half_t)w4_acc_read<base + 3>();
      *(half4_t*)(stg + (16 * ih + r16) * W4_EPI_STRIDE + (16 * j + 4 * kg) * 2) = h;
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = it * 4 + (lane >> 4);
      const u32x4_t v = *(const u32x4_t*)(stg + row * W4_EPI_STRIDE + (lane & 15) * 16);
      *(u32x4_t*)(cw + (size_t)(q * 32 + row) * N + (lane & 15) * 8) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  });
}

template <bool B_KN>
__global__ __launch_bounds__(256) void hgemm_w4x_kernel(const half_t* __restrict__ A, const half_t* __restrict__ B,
                                                       half_t* __restrict__ C, int M, int N, int K, int tiles_m,
                                                       int tiles_n, int panel_w) {
  static_assert(!B_KN, "hgemm_w4x_kernel: TN (B stored [N][K]) only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = wave_id();
  const int wr = wave >> 1, wc = wave & 1;

  const TileCoord tc = block_tile(blockIdx.x, gridDim.x, tiles_m, tiles_n, panel_w);
  const int m0 = tc.tm * BM, n0 = tc.tn * BN;

  // ---- DMA: identical to hgemm_w4b_kernel<false, true, false> (pieces g = 0..7 A, 8..15 B; tile clamped past the end)
  W4Src<B_KN> src;
  w4_src_init<B_KN>(src, A, B, m0, n0, N, K, wave, lane);
  const int KT = K / BK;
  const buf_rsrc_t ra = make_rsrc(src.ua), rb = make_rsrc(src.ub);
  auto piece = [&](int g, int t, char* slot) {
    const int te = t < KT ? t : KT - 1;
    if (g < 8) {
      blds16(ra, src.a_off[g & 1], (unsigned)((size_t)g * src.a_blk + (size_t)te * src.a_kt), slot + (wave * 8 + g) * 1024);
    } else {
      const int p = g - 8;
      blds16(rb, src.b_off[p & 1], (unsigned)((size_t)p * src.a_blk + (size_t)te * src.b_kt), slot + (wave * 8 + p) * 1024);
    }
  };
  auto a_slot = [&](int t) -> char* { return smem + (t & 1) * TILE_BYTES; };
  auto b_slot = [&](int bi) -> char* { return smem + (2 + bi) * TILE_BYTES; };   // bi = t % 3, tracked by the caller

  W4xFrag fr;
  w4x_frag_init(fr, wr, wc, lane);
  auto read_a = [&](const char* slot, int ks, int i) -> half8_t { return *(const half8_t*)(slot + fr.a_ad[ks] + i * 2048); };
  auto read_b = [&](const char* slot, int ks, int j) -> half8_t { return *(const half8_t*)(slot + fr.b_ad[ks] + j * 2048); };

  static_for<256>([&](auto r) { w4_acc_zero<decltype(r)::value>(); });

  // prologue: B(0) A(0) B(1) A(1); tile 0 landed; fragments of step 0 (k-step 0: A rows 0..3, all of B) in registers
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

  half8_t xa[2][4], xb[2][8];
#pragma unroll
  for (int i = 0; i < 4; ++i) xa[0][i] = read_a(a_slot(0), 0, i);
#pragma unroll
  for (int j = 0; j < 8; ++j) xb[0][j] = read_b(b_slot(0), 0, j);

  // one step = 32 MFMAs: A fragments xa[S & 1] (rows 4(S & 1) .. +3 of k-step S >> 1) x B fragments xb[S >> 1];
  // chunks 0..3 read the next step's A fragments from rsa, chunks 4..11 of an odd step the next k-step's B fragments from
  // rsb; DMA: NP pieces g0.. of tile t2 into wslot, spread over the 16 chunks
  auto step = [&](auto sc, const char* rsa, const char* rsb, auto npc, int g0, int t2, char* wslot) {
    constexpr int S = decltype(sc)::value, ks = S >> 1, hf = S & 1;
    constexpr int NS = (S + 1) & 3, nks = NS >> 1, nh = NS & 1;
    constexpr int NP = decltype(npc)::value;
    static_for<16>([&](auto cc) {
      constexpr int c = decltype(cc)::value, ii = c >> 2, jp = c & 3;
      w4x_mfma2<8 * (4 * hf + ii) + 2 * jp, 8 * (4 * hf + ii) + 2 * jp + 1>(xb[ks][2 * jp], xb[ks][2 * jp + 1], xa[hf][ii]);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (c < 4) xa[hf ^ 1][c] = read_a(rsa, nks, 4 * nh + c);
      if constexpr (hf == 1 && c >= 4 && c < 12) xb[ks ^ 1][c - 4] = read_b(rsb, nks, c - 4);
      if constexpr (NP == 8 && (c & 1) == 1) piece(g0 + (c >> 1), t2, wslot);
      if constexpr (NP == 4 && (c & 3) == 3) piece(g0 + (c >> 2), t2, wslot);
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  using S3 = std::integral_constant<int, 3>;
  using P0 = std::integral_constant<int, 0>;
  using P4 = std::integral_constant<int, 4>;
  using P8 = std::integral_constant<int, 8>;

  int b0 = 0, b1 = 1, b2 = 2;   // B slot indices of tiles kt, kt+1, kt+2 (rotating, kt % 3)
  for (int kt = 0; kt < KT; ++kt) {
    const char* ca = a_slot(kt);
    const char* cbs = b_slot(b0);
    step(S0{}, ca, cbs, P0{}, 0, 0, nullptr);
    step(S1{}, ca, cbs, P4{}, 8, kt + 2, b_slot(b2));
    step(S2{}, ca, cbs, P4{}, 12, kt + 2, b_slot(b2));
    // every read of tile kt is issued; A(kt+1), B(kt+1) must have landed (the 8 B pieces of tile kt+2 stay in flight)
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    pp_barrier();
    step(S3{}, a_slot(kt + 1), b_slot(b1), P8{}, 0, kt + 2, a_slot(kt));
    const int t = b0;
    b0 = b1;
    b1 = b2;
    b2 = t;
  }
  LC_VMCNT(0);
  w4x_epilogue(smem, C, N, m0, n0, wave, wr, wc, lane);
}

}  // namespace lc
