// gemm_fp8.hip — fp8 (OCP e4m3fn) GEMM for gfx950: C[M,N] (fp16) = alpha · A8[M,K] · B8ᵀ,  B8 stored [N,K].
//
// BASELINE config 5 ("fp8 MFMA HGEMM M=N=K=16384").  The reference has no fp8 GEMM (its wrappers hard-require
// kHalf) — this is an EXTENSION of the path, parity is defined against the fp64 oracle on the
// decoded e4m3 values.  Same 256x256 workgroup tile / 8 wave64 / two-phase ping-pong schedule, LDS ring,
// swizzle and DMA plan as hgemm_pingpong2_kernel — byte for byte: a 128-byte LDS row now holds 128 k-values,
// so a K tile is 128 deep and every ds_read_b128 fragment feeds TWO v_mfma_f32_32x32x16_fp8_fp8 (low / high
// 8 bytes; both operands are split the same way, and MFMA contracts over matching (lane-half, slot) pairs, so
// any consistent k assignment is exact).  32 MFMAs per phase instead of 16: the barrier / load-section
// overhead per MFMA halves.  Non-scaled fp8 MFMA runs at the fp16 rate (2.5 PF dense).
// MX = true: the same data through v_mfma_scale_f32_32x32x64_f8f6f4 with both block scales = 2^0 (E8M0 0x7F): one
// instruction contracts 64 k-values (two 16-byte fragments per operand), at twice the MAC rate of the non-scaled
// form; e4m3 x e4m3 products are exact in fp32 either way, only the accumulation order differs.
#pragma once
#include "hgemm_pingpong.hip"
#include "hgemm_w4.hip"

namespace lc {

constexpr int BK8 = 128;  // k elements (= bytes) per K tile

typedef long i64x2_t __attribute__((ext_vector_type(2)));

LC_DEVINL f32x16_t mfma32_fp8(long a, long b, f32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a, b, c, 0, 0, 0);
}

struct PPSrc8 {
  const uint8_t* a[2][2];
  const uint8_t* b[2][2];
  int a_lds[2][2];
  int b_lds[2][2];
};

LC_DEVINL void pp_src8_init(PPSrc8& s, const uint8_t* A, const uint8_t* B, int m0, int n0, int K, int wave,
                            int lane) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = wave * 2 + i;
      {
        const int blk = 16 * (q >> 3) + 8 * h + (q & 7);
        const int row = blk * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        s.a[h][i] = A + (size_t)(m0 + row) * K + c * 16;
        s.a_lds[h][i] = blk * 1024;
      }
      {
        const int blk = 8 * (q >> 2) + 4 * h + (q & 3);
        const int row = blk * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        s.b[h][i] = B + (size_t)(n0 + row) * K + c * 16;
        s.b_lds[h][i] = TILE_BYTES + blk * 1024;
      }
    }
  }
}

typedef int i32x8_t __attribute__((ext_vector_type(8)));
LC_DEVINL i32x8_t cat8(half8_t a, half8_t b) {
  typedef int i32x4_t __attribute__((ext_vector_type(4)));
  return __builtin_shufflevector(__builtin_bit_cast(i32x4_t, a), __builtin_bit_cast(i32x4_t, b), 0, 1, 2, 3, 4, 5, 6, 7);
}
LC_DEVINL f32x16_t mfma32_fp8_mx(i32x8_t a, i32x8_t b, f32x16_t c) {   // formats 0 = e4m3, scales 1.0
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}

// MX form of the cluster: 8 MFMAs of K = 64 (fragments ks = 2k', 2k'+1 of both operands form one 32-byte operand,
// read straight into the two halves of an 8-register tuple: no concatenation copies)
typedef int i32x4_t __attribute__((ext_vector_type(4)));
LC_DEVINL void pp8_read_a(const char* slot, const PPFrag<false>& f, int mh, i32x8_t (&af)[2][2]) {
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int kp = 0; kp < 2; ++kp) {
      af[m][kp].lo = *(const i32x4_t*)(slot + ((f.a0 ^ ((2 * kp) * 32)) + (mh * 2 + m) * 4096));
      af[m][kp].hi = *(const i32x4_t*)(slot + ((f.a0 ^ ((2 * kp + 1) * 32)) + (mh * 2 + m) * 4096));
    }
}
LC_DEVINL void pp8_read_b(const char* slot, const PPFrag<false>& f, int nh, i32x8_t (&bf)[2]) {
#pragma unroll
  for (int kp = 0; kp < 2; ++kp) {
    bf[kp].lo = *(const i32x4_t*)(slot + ((f.b0 ^ ((2 * kp) * 32)) + nh * 4096));
    bf[kp].hi = *(const i32x4_t*)(slot + ((f.b0 ^ ((2 * kp + 1) * 32)) + nh * 4096));
  }
}
template <int NG, typename IssueFn>
LC_DEVINL void pp8_cluster_mx(f32x16_t (&acc)[4][2], int mh, const i32x8_t (&af)[2][2], const i32x8_t (&b0f)[2],
                              const i32x8_t (&b1f)[2], IssueFn issue) {
  __builtin_amdgcn_s_setprio(1);
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const int kp = g >> 2, nh = (g >> 1) & 1, m = g & 1;
    const i32x8_t bv = nh ? b1f[kp] : b0f[kp];
    const i32x8_t av = af[m][kp];
    acc[mh * 2 + m][nh] = mfma32_fp8_mx(bv, av, acc[mh * 2 + m][nh]);
    if (g < NG) issue(g);
    __builtin_amdgcn_sched_barrier(0);
  }
  __builtin_amdgcn_s_setprio(0);
}

template <int NG, typename IssueFn>
LC_DEVINL void pp8_cluster(f32x16_t (&acc)[4][2], int mh, const half8_t (&af)[2][4], const half8_t (&b0f)[4],
                           const half8_t (&b1f)[4], IssueFn issue) {
  __builtin_amdgcn_s_setprio(1);
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const int ks = g >> 1, nh = g & 1;
    const i64x2_t bv = __builtin_bit_cast(i64x2_t, nh ? b1f[ks] : b0f[ks]);
    const i64x2_t a0 = __builtin_bit_cast(i64x2_t, af[0][ks]);
    const i64x2_t a1 = __builtin_bit_cast(i64x2_t, af[1][ks]);
    acc[mh * 2 + 0][nh] = mfma32_fp8(bv[0], a0[0], acc[mh * 2 + 0][nh]);
    acc[mh * 2 + 1][nh] = mfma32_fp8(bv[0], a1[0], acc[mh * 2 + 1][nh]);
    acc[mh * 2 + 0][nh] = mfma32_fp8(bv[1], a0[1], acc[mh * 2 + 0][nh]);
    acc[mh * 2 + 1][nh] = mfma32_fp8(bv[1], a1[1], acc[mh * 2 + 1][nh]);
    if (g < NG) issue(g);
    __builtin_amdgcn_sched_barrier(0);
  }
  __builtin_amdgcn_s_setprio(0);
}

template <bool MX>
__global__ __launch_bounds__(512, 2) void gemm_fp8_pingpong2_kernel(const uint8_t* __restrict__ A,
                                                                    const uint8_t* __restrict__ B,
                                                                    half_t* __restrict__ C, int M, int N, int K,
                                                                    float alpha, int tiles_m, int tiles_n,
                                                                    int panel_w) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = wave_id();
  const int wr = wave >> 2, wc = wave & 3;

  const TileCoord tc = block_tile(blockIdx.x, gridDim.x, tiles_m, tiles_n, panel_w);
  const int m0 = tc.tm * BM, n0 = tc.tn * BN;

  PPSrc8 src;
  pp_src8_init(src, A, B, m0, n0, K, wave, lane);
  PPFrag<false> fr;
  pp_frag_init<false>(fr, wr, wc, lane);

  f32x16_t acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int KT = K / BK8;
  auto piece = [&](int is_b, int h, int i, int t) {
    const int te = t < KT ? t : KT - 1;
    char* slot = smem + (t & 1) * SLOT_BYTES;
    if (is_b)
      glds16(src.b[h][i] + (size_t)te * BK8, slot + src.b_lds[h][i]);
    else
      glds16(src.a[h][i] + (size_t)te * BK8, slot + src.a_lds[h][i]);
  };
  auto issue_ab0 = [&](int g, int t) { piece(g < 4, g < 4 ? (g >> 1) : 0, g & 1, t); };

#pragma unroll
  for (int g = 0; g < 6; ++g) issue_ab0(g, 0);
  piece(0, 1, 0, 0);
  piece(0, 1, 1, 0);
#pragma unroll
  for (int g = 0; g < 6; ++g) issue_ab0(g, 1);
  LC_VMCNT(8);
  pp_barrier();
  if (wr == 1) pp_barrier();

  half8_t af[2][4], b0f[4], b1f[4];   // 16-byte fragments (16 fp8 values each)
  i32x8_t af2[2][2], b0f2[2], b1f2[2];   // MX: 32-byte operands
  for (int kt = 0; kt < KT; ++kt) {
    const char* cur = smem + (kt & 1) * SLOT_BYTES;
    if constexpr (MX) {
      pp8_read_b(cur, fr, 0, b0f2);
      pp8_read_a(cur, fr, 0, af2);
      pp8_read_b(cur, fr, 1, b1f2);
    } else {
      pp_read_b<false>(cur, fr, 0, b0f);
      pp_read_a<false>(cur, fr, 0, af);
      pp_read_b<false>(cur, fr, 1, b1f);
    }
    LC_VMCNT(6);
    pp_barrier();
    if constexpr (MX) pp8_cluster_mx<2>(acc, 0, af2, b0f2, b1f2, [&](int g) { piece(0, 1, g, kt + 1); });
    else pp8_cluster<2>(acc, 0, af, b0f, b1f, [&](int g) { piece(0, 1, g, kt + 1); });
    pp_barrier();
    if constexpr (MX) pp8_read_a(cur, fr, 1, af2); else pp_read_a<false>(cur, fr, 1, af);
    LC_VMCNT(2);
    pp_barrier();
    if constexpr (MX) pp8_cluster_mx<6>(acc, 1, af2, b0f2, b1f2, [&](int g) { issue_ab0(g, kt + 2); });
    else pp8_cluster<6>(acc, 1, af, b0f, b1f, [&](int g) { issue_ab0(g, kt + 2); });
    pp_barrier();
  }
  if (wr == 0) pp_barrier();
  LC_VMCNT(0);
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] *= alpha;
  pp_epilogue(smem, acc, C, N, m0, n0, wave, wr, wc, lane);
}


// ------------------------------------------------------------------------------------------------
// gemm_fp8_w4_kernel — the fp8 GEMM on the structure of hgemm_w4b_kernel<TN, BUF> (four waves, 128x128 wave tiles,
// 4x4 accumulators in literal AGPRs, A ring of 2 + B ring of 3 K tiles, buffer_load..lds DMA, one barrier per K tile)
// with the MX-scaled K = 64 MFMA (unit block scales).  A K tile is 128 k = 128 bytes per row = TWO k-steps of 64; a
// k-step is 16 MFMAs (64 cycles each) from 4 + 4 operands of 32 bytes per lane (two ds_read_b128 each, read straight
// into the halves of an 8-register tuple), double-buffered:
//   step 0: MFMAs on k-step 0 | reads of k-step 1            | DMA B(t+2): 8 pieces
//           s_waitcnt vmcnt(8) lgkmcnt(0); s_barrier          (A(t+1), B(t+1) landed; 8 B pieces stay in flight)
//   step 1: MFMAs on k-step 1 | reads of k-step 0 of tile t+1 | DMA A(t+2): 8 pieces
template <int IDX>
LC_DEVINL void w4_mfma_fp8mx(i32x8_t a, i32x8_t b, int scale) {   // a[16 IDX .. +15] += a x b   (formats e4m3, scales 2^0)
  asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 a[%3:%4], %0, %1, a[%3:%4], %2, %2 op_sel_hi:[0,0,0]"
               :: "v"(a), "v"(b), "v"(scale), "n"(IDX * 16), "n"(IDX * 16 + 15) : LC_AGPR_ALL);
}

__global__ __launch_bounds__(256) void gemm_fp8_w4_kernel(const uint8_t* __restrict__ A, const uint8_t* __restrict__ B,
                                                          half_t* __restrict__ C, int M, int N, int K, float alpha,
                                                          int tiles_m, int tiles_n, int panel_w, int stagger) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = wave_id();
  const int wr = wave >> 1, wc = wave & 1;
  const int l32 = lane & 31, hi = lane >> 5;

  const TileCoord tc = block_tile(blockIdx.x, gridDim.x, tiles_m, tiles_n, panel_w);
  const int m0 = tc.tm * BM, n0 = tc.tn * BN;
  const int KT = K / BK8;

  // DMA: piece p of this wave = the 8-row block 4 p + wave of the A / B tile (rows 32 p + 8 wave .. + 8: the four waves' concurrent
  // requests cover 32 consecutive rows — hgemm_w4y.hip's piece map); lane -> row lane>>3, 16-byte slot lane&7 (swizzle as
  // hgemm_w4: key ((row >> 1) & 7) = (lane >> 4) & 3 | (block & 1) << 2, block & 1 = wave & 1)
  const unsigned a_off = (unsigned)(lane >> 3) * (unsigned)K + (unsigned)(((lane & 7) ^ (((lane >> 4) & 3) | ((wave & 1) << 2))) * 16);
  const buf_rsrc_t ra = make_rsrc(A + (size_t)(m0 + wave * 8) * K);
  const buf_rsrc_t rb = make_rsrc(B + (size_t)(n0 + wave * 8) * K);
  const unsigned blk_bytes = 32u * (unsigned)K;
  // K-loop stagger (hgemm_w4y.hip; lc_tune_set "hgemm_stagger"): the workgroup walks the K tiles stg, stg + 1, ..., wrapping
  int stg;
  {
    const int cx = stagger & 15, cm = (stagger >> 4) & 15, cn = (stagger >> 8) & 15, step = (stagger >> 12) & 0xff, mask = (stagger >> 20) & 0x7f;
    const int idx = cx * __builtin_amdgcn_readfirstlane((int)(blockIdx.x & 7)) + cm * __builtin_amdgcn_readfirstlane(tc.tm) +
                    cn * __builtin_amdgcn_readfirstlane(tc.tn);
    stg = __builtin_amdgcn_readfirstlane((int)((unsigned)((idx & mask) * step) % (unsigned)KT));
  }
  auto piece = [&](int g, int t, char* slot) {   // g < 8: A pieces, else B pieces; clamped past the end
    int te = (t < KT ? t : KT - 1) + stg;
    if (te >= KT) te -= KT;
    const int p = g & 7;
    blds16(g < 8 ? ra : rb, a_off, (unsigned)p * blk_bytes + (unsigned)te * BK8, slot + p * 4096 + wave * 1024);
  };
  auto a_slot = [&](int t) -> char* { return smem + (t & 1) * TILE_BYTES; };
  auto b_slot = [&](int bi) -> char* { return smem + (2 + bi) * TILE_BYTES; };

  // fragment addresses: 16-byte chunk c = 2ks + hi of row r at slot c ^ ((r>>1)&7); ks = 0..3 (16 bytes of k each)
  const int swz = (lane >> 1) & 7;
  int a_ad[4], b_ad[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    a_ad[ks] = (wr * 128 + l32) * 128 + (((2 * ks + hi) ^ swz) * 16);
    b_ad[ks] = (wc * 128 + l32) * 128 + (((2 * ks + hi) ^ swz) * 16);
  }
  auto read_op = [&](const char* slot, const int (&ad)[4], int kp, int i) -> i32x8_t {   // k-step kp, 32-row block i
    i32x8_t v;
    v.lo = *(const i32x4_t*)(slot + ad[2 * kp] + i * 4096);
    v.hi = *(const i32x4_t*)(slot + ad[2 * kp + 1] + i * 4096);
    return v;
  };

  static_for<256>([&](auto r) { w4_acc_zero<decltype(r)::value>(); });
  int scale = 0x7f7f7f7f;
  asm volatile("" : "+v"(scale));

  // prologue: B(0) A(0) B(1) A(1)
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

  i32x8_t af[2][4], bf[2][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    af[0][i] = read_op(a_slot(0), a_ad, 0, i);
    bf[0][i] = read_op(b_slot(0), b_ad, 0, i);
  }

  auto step = [&](auto cbc, const char* ra_, const char* rb_, int rkp, int g0, int t2, char* wslot) {
    constexpr int cb = decltype(cbc)::value;
    static_for<8>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      constexpr int i = c >> 1, j0 = 2 * (c & 1);
      w4_mfma_fp8mx<4 * i + j0>(bf[cb][j0], af[cb][i], scale);
      w4_mfma_fp8mx<4 * i + j0 + 1>(bf[cb][j0 + 1], af[cb][i], scale);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (c < 4) {
        af[cb ^ 1][c] = read_op(ra_, a_ad, rkp, c);
        bf[cb ^ 1][c] = read_op(rb_, b_ad, rkp, c);
      }
      piece(g0 + c, t2, wslot);
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  int b0 = 0, b1 = 1, b2 = 2;
  for (int kt = 0; kt < KT; ++kt) {
    step(I0{}, a_slot(kt), b_slot(b0), 1, 8, kt + 2, b_slot(b2));
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    pp_barrier();
    step(I1{}, a_slot(kt + 1), b_slot(b1), 0, 0, kt + 2, a_slot(kt));
    const int t = b0;
    b0 = b1;
    b1 = b2;
    b2 = t;
  }
  LC_VMCNT(0);

  // epilogue shared with hgemm_w4 (alpha applied in fp32); a K = 64 MFMA is 16 passes: two extra drains
  w4_mfma_drain();
  w4_mfma_drain();
  w4_epilogue<true>(smem, C, N, m0, n0, wave, wr, wc, lane, alpha);
}

}  // namespace lc
