// attn_w4n.hip — FlashAttention-2 forward, D = 128: the merged-phase 4-wave kernel of attn_w4m.hip with
// v_mfma_f32_16x16x32_f16 instead of v_mfma_f32_32x32x16_f16 (lc_tune_set "attn_nw" = 512; round 2).
//
// Same semantics / entry points as attn_fwd.hip (reference: kernels/flash-attn/mma/basic/
// flash_attn_mma_split_q.cu:55-699, flash_attn_mma_share_qkv.cu:46-769).
//
// Why: the merged-phase kernel runs at the board's 1400 W cap on random data (1.19-1.23 PFLOP/s at 1.7-1.85 GHz; the same
// binary on zero-filled inputs 1.64 at 2.39 GHz), and an MFMA-only stream at that cap sustains 14 % more FLOP/s with the
// 16x16x32 form than with 32x32x16 (half the accumulator registers moved per FLOP; DESIGN.md §4.10,
// profiles/r2_power_probe.log).  Everything else is attn_w4m's design, re-derived for the 16-wide shapes:
//   * wave = 64 query rows = 4 query blocks qb of 16; half-tile j = 32 KV rows = 2 kv blocks kvb of 16;
//   * Sᵀ block (kvb, qb) = K fragment (kvb, ds) x Q~ fragment (qb, ds), ds = 0..3 (32 d each): lane holds
//     S[q = 16 qb + (l & 15)][kv = 16 kvb + 4 (l >> 4) + r], r = 0..3 (4 registers per block, 8 blocks = 32 registers);
//     the first MFMA of a block takes C = −m (4-register tuple), so the block comes out as s − m;
//   * the Pᵀ operand of P·V is lane-local: its k slot 8 g + e (g = l >> 4) is DEFINED as kv = 16 (e >> 2) + 4 g + (e & 3),
//     i.e. P(qb) = pack(S(0, qb)[0..3], S(1, qb)[0..3]); the Vᵀ operand follows the same slot order with two
//     ds_read_b64_tr_b16 per fragment (kv rows 4 g.. of kv block 0, then of kv block 1);
//   * Oᵀ block (db, qb) = a[4 (4 db + qb) ..]: lane holds O[q = 16 qb + (l & 15)][d = 16 db + 4 g + r] (128 AGPRs);
//   * a phase = 64 MFMAs, Q·Kᵀ(j+1) (32: ds outer, block inner — a block is touched every 16th slot) alternating with
//     P·V(j−1) (32: db outer, qb inner), softmax(j) in their shadow (per 4 slots: 2 v_exp, 2 v_add, 1 v_cvt_pk, each v_exp behind a
//     plain VALU); LDS reads: slots 0..7 the K(j+2) fragments (-> AGPR) and Vᵀ(j−1) blocks db 4..7, slots 32..39 Vᵀ(j) blocks
//     db 0..3 (their registers were last read by the P·V MFMA of slot 31);
//   * V tile swizzle: 32-B column pair p of row r at pair slot p ^ (((r & 3) << 1) | ((r >> 2) & 1)) — the 64-B-unit
//     swizzle of attn_w4m is 2-way conflicted for these transpose reads, whose 32-lane groups span rows 4 apart
//     (tests/test_layouts.py); K tile swizzle, ring, DMA schedule, barrier, overflow slow path, Os2g epilogue: as attn_w4m.
// Register plan (literal AGPRs): a[0:127] Oᵀ; a[128:191] two K half-tile buffers (fragment (kvb, ds) at +4 (4 kvb + ds));
// a[192:255] Q~ fragments (qb, ds) at +4 (4 qb + ds).
#pragma once
#include "attn_w4m.hip"

namespace lc {

// introspection (lc_attn_slowpath_stats): how often the overflow slow path ran — [0] executions, [1] sum of half-tile indices
// j, [2] how many of them saw a non-finite row sum, [3] bit pattern of the last offending row sum.  One atomic per execution
// of a path that N(0,1) inputs never take.
// (one counter block per translation unit that instantiates these kernels: LC_AN_SLOWPATH_SYM names it)
#ifndef LC_AN_SLOWPATH_SYM
#define LC_AN_SLOWPATH_SYM g_an_slowpath
#endif
__device__ unsigned int LC_AN_SLOWPATH_SYM[4];

constexpr int AN_O = 0, AN_K = 128, AN_Q = 192;

// one MFMA slot = ONE asm statement (see am_slot): KIND 0 Sᵀ block = K frag x Q~ frag + C; 1 accumulate; 2 Oᵀ block +=
// Vᵀ frag (VGPR) x Pᵀ frag (VGPR); 3 none.  RD bit 0: + ds_read_b64_tr_b16 vout <- [vaddr + VOF]; bit 1: + ds_read_b128
// a[KR:+3] <- [kaddr + KOF].
template <int KIND, int RD, int R0, int R1, int VOF, int KR, int KOF>
LC_DEVINL void an_slot(f32x4_t& sblk, const f32x4_t& cblk, half8_t vfrag, half8_t pfrag, half4_t& vout, uint32_t vaddr,
                       uint32_t kaddr) {
  uint32_t vdummy;
  if constexpr (KIND == 0) {
    AM_SLOT_BODY("v_mfma_f32_16x16x32_f16 %[s], a[%[r0]:%[r0e]], a[%[r1]:%[r1e]], %[c]", [s] "=&v"(sblk) AM_COMMA,
                 [c] "v"(cblk) AM_COMMA [r0] "n"(R0) AM_COMMA [r0e] "n"(R0 + 3) AM_COMMA [r1] "n"(R1) AM_COMMA [r1e] "n"(R1 + 3) AM_COMMA)
  } else if constexpr (KIND == 1) {
    AM_SLOT_BODY("v_mfma_f32_16x16x32_f16 %[s], a[%[r0]:%[r0e]], a[%[r1]:%[r1e]], %[s]", [s] "+v"(sblk) AM_COMMA,
                 [r0] "n"(R0) AM_COMMA [r0e] "n"(R0 + 3) AM_COMMA [r1] "n"(R1) AM_COMMA [r1e] "n"(R1 + 3) AM_COMMA)
  } else if constexpr (KIND == 2) {
    AM_SLOT_BODY("v_mfma_f32_16x16x32_f16 a[%[r0]:%[r0e]], %[vf], %[pf], a[%[r0]:%[r0e]]", ,
                 [vf] "v"(vfrag) AM_COMMA [pf] "v"(pfrag) AM_COMMA [r0] "n"(R0) AM_COMMA [r0e] "n"(R0 + 3) AM_COMMA)
  } else {
    AM_SLOT_BODY("", , )
  }
}
template <int KREG, int QREG>
LC_DEVINL void an_qk_zero(f32x4_t& s) {
  asm volatile("v_mfma_f32_16x16x32_f16 %0, a[%1:%2], a[%3:%4], 0"
               : "=&v"(s) : "n"(KREG), "n"(KREG + 3), "n"(QREG), "n"(QREG + 3) : LC_AGPR_ALL);
}
template <int KREG, int QREG>
LC_DEVINL void an_qk(f32x4_t& s) {
  asm volatile("v_mfma_f32_16x16x32_f16 %0, a[%1:%2], a[%3:%4], %0"
               : "+v"(s) : "n"(KREG), "n"(KREG + 3), "n"(QREG), "n"(QREG + 3) : LC_AGPR_ALL);
}
template <int OACC>
LC_DEVINL void an_pv(half8_t v, half8_t p) {
  asm volatile("v_mfma_f32_16x16x32_f16 a[%2:%3], %0, %1, a[%2:%3]"
               :: "v"(v), "v"(p), "n"(OACC), "n"(OACC + 3) : LC_AGPR_ALL);
}
// reductions over the four 16-lane groups (a query row's kv columns are spread over lanes l, l^16, l^32, l^48); only
// the prologue, the slow path and the epilogue use them
LC_DEVINL float an_x4_max(float x) {
  x = fmaxf(x, __shfl_xor(x, 16));
  return fmaxf(x, __shfl_xor(x, 32));
}
LC_DEVINL float an_x4_sum(float x) {
  x += __shfl_xor(x, 16);
  return x + __shfl_xor(x, 32);
}

template <int D>
__global__ __launch_bounds__(256) void attn_fwd_w4n_kernel(
    const half_t* __restrict__ Q, const half_t* __restrict__ K, const half_t* __restrict__ V,
    half_t* __restrict__ O, int N, int nqb, float sl2) {
  static_assert(D == 128, "w4n attention kernel: D = 128 only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = wave_id();
  const int g4 = lane >> 4, l16 = lane & 15;

  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const size_t bh = id / nqb;
  const int q0 = (id - (int)bh * nqb) * 256 + wave * 64;
  const half_t* Qb = Q + bh * (size_t)N * D;
  const char* Kb = (const char*)(K + bh * (size_t)N * D);
  const char* Vb = (const char*)(V + bh * (size_t)N * D);
  half_t* Ob = O + bh * (size_t)N * D;
  const int T = N / KVB;
  const uint32_t smem32 = lds_addr32(smem);

  // ---- LDS-DMA: piece p = 4 rows x 256 B; this wave stages pieces wave + 4i (i = 0..3) of K and of V
  const int r4 = lane >> 4, cs = lane & 15;
  const unsigned k_off = (unsigned)(r4 * 256 + ((cs ^ (4 * wave + r4)) * 16));   // chunk ^ (row & 15), row & 15 = 4 (p & 3) + r4
  // V: pair slot (cs >> 1) holds logical pair (cs >> 1) ^ key(row), key = ((row & 3) << 1) | ((row >> 2) & 1);
  // row = 4 p + r4 -> row & 3 = r4, (row >> 2) & 1 = p & 1 = wave & 1
  const unsigned v_off = (unsigned)(r4 * 256 + (((((cs >> 1) ^ ((r4 << 1) | (wave & 1))) << 1) | (cs & 1)) * 16));
  const buf_rsrc_t rk = make_rsrc(Kb), rv = make_rsrc(Vb);
  auto issue_piece = [&](int i, int t) {   // i = 0..7: K pieces, then V pieces; tile t (clamped) -> ring slot t & 3
    const int te = t < T ? t : T - 1;
    char* slot = smem + (t & 3) * AM_SLOT;
    const int p = wave + 4 * (i & 3);
    const unsigned so = (unsigned)te * AM_TILE + (unsigned)p * 1024u;
    if (i < 4)
      blds16(rk, k_off, so, slot + p * 1024);
    else
      blds16(rv, v_off, so, slot + AM_TILE + p * 1024);
  };
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < 8; ++i) issue_piece(i, t);

  // ---- Q~ = fp16(Q * scale*log2e) -> AGPRs: lane holds Q[q0 + 16 qb + l16][32 ds + 8 g4 .. +8]
  static_for<16>([&](auto ic) {
    constexpr int i = decltype(ic)::value, qb = i >> 2, ds = i & 3;
    const half8_t q = *(const half8_t*)(Qb + (size_t)(q0 + 16 * qb + l16) * D + 32 * ds + 8 * g4);
    half8_t qs;
#pragma unroll
    for (int e = 0; e < 8; ++e) qs[e] = (half_t)((float)q[e] * sl2);
    const u32x4_t w = __builtin_bit_cast(u32x4_t, qs);
    am_acc_write<AN_Q + 4 * i + 0>(w[0]);
    am_acc_write<AN_Q + 4 * i + 1>(w[1]);
    am_acc_write<AN_Q + 4 * i + 2>(w[2]);
    am_acc_write<AN_Q + 4 * i + 3>(w[3]);
  });
  static_for<128>([&](auto r) { am_acc_zero<decltype(r)::value>(); });

  // ---- fragment read offsets inside a ring slot
  uint32_t kx[4];   // K: row l16 (+16 kvb, +32 per half-tile: immediates), 16-B chunk (4 ds + g4) ^ (row & 15)
#pragma unroll
  for (int ds = 0; ds < 4; ++ds) kx[ds] = (uint32_t)(l16 * 256 + (((4 * ds + g4) ^ l16) * 16));
  uint32_t vx[4];   // Vᵀ transpose reads: kv row 4 g4 + (l16 >> 2) (+16 x, +32 per half-tile: immediates), pair db = 2u + (db & 1)
#pragma unroll
  for (int u = 0; u < 4; ++u)
    vx[u] = (uint32_t)(AM_TILE + (4 * g4 + (l16 >> 2)) * 256 + (((2 * u) ^ (((l16 >> 2) << 1) | (g4 & 1))) * 32) + 8 * (l16 & 3));
  // (pair 2u + 1 sits at pair slot (2u ^ key) ^ 1: the XOR of bit 0 is not an immediate offset -> handled below)
  const uint32_t vodd = (uint32_t)((g4 & 1) ? -32 : 32);   // byte offset from pair 2u's slot to pair 2u + 1's: key bit 0 = g4 & 1

  uint32_t ka[4], vc[4], vp[4];   // this tile period's LDS addresses: K(t+1) fragments, Vᵀ of tile t / tile t−1
  auto set_tile_addrs = [&](int t) {
    const uint32_t sb_cur = smem32 + (uint32_t)((t & 3) * AM_SLOT), sb_nxt = smem32 + (uint32_t)(((t + 1) & 3) * AM_SLOT);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      vp[u] = vc[u];
      vc[u] = vx[u] + sb_cur;
    }
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) ka[ds] = kx[ds] + sb_nxt;
  };
#pragma unroll
  for (int u = 0; u < 4; ++u) vc[u] = vx[u] + smem32;

  f32x4_t sA[2][4], sB[2][4];   // Sᵀ blocks [kvb][qb] of the half-tile being exponentiated / being accumulated
  f32x4_t negm[4];              // C operand of the first d-step: 4 x (−m) per query block
  half8_t pA[4], pB[4];         // P fragments [qb]
  half4_t vlo[8], vhi[8];       // Vᵀ fragments [db]: lo = kv block 0 rows, hi = kv block 1 rows (set A = db 0..3, set B = 4..7)
  float l_run[4] = {0.f, 0.f, 0.f, 0.f};

  auto read_k_all = [&](auto bufc, uint32_t sbase, auto hc) {   // (prologue only) one K half-tile -> AGPR buffer BUF
    constexpr int BUF = decltype(bufc)::value, H = decltype(hc)::value;
    static_for<8>([&](auto cc) {
      constexpr int c = decltype(cc)::value, kvb = c >> 2, ds = c & 3;
      am_read_k<AN_K + 32 * BUF + 4 * c, H * 8192 + kvb * 4096>(kx[ds] + sbase);
    });
  };
  // Vᵀ fragment db of half-tile H' of the tile at `va`: two transpose reads (kv block 0 / 1 rows)
  auto vaddr_of = [&](const uint32_t (&va)[4], int db) -> uint32_t { return va[db >> 1] + ((db & 1) ? vodd : 0u); };

  // ---- prologue: tiles 0, 1 landed; K(0), K(1) -> AGPR buffers 0, 1; Sᵀ(0), its row max, S − m, −m tuples
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  raw_barrier();
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  read_k_all(I0{}, smem32, I0{});
  read_k_all(I1{}, smem32, I1{});
  am_lgkm0();
  static_for<32>([&](auto ic) {
    constexpr int i = decltype(ic)::value, ds = i >> 3, kvb = (i >> 2) & 1, qb = i & 3;
    if constexpr (ds == 0) an_qk_zero<AN_K + 4 * (4 * kvb + ds), AN_Q + 4 * (4 * qb + ds)>(sA[kvb][qb]);
    else an_qk<AN_K + 4 * (4 * kvb + ds), AN_Q + 4 * (4 * qb + ds)>(sA[kvb][qb]);
  });
  am_drain(sA);
#pragma unroll
  for (int qb = 0; qb < 4; ++qb) {
    float mx = sA[0][qb][0];
#pragma unroll
    for (int kvb = 0; kvb < 2; ++kvb)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sA[kvb][qb][r]);
    mx = an_x4_max(mx);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sA[0][qb][r] -= mx;
      sA[1][qb][r] -= mx;
      negm[qb][r] = -mx;
    }
  }

  // ---- one merged phase.  H = half-tile parity (j = 2t + H); F: 1 = P·V(j−1) exists, 2 = Q·Kᵀ(j+1) exists,
  // 4 = K(j+2) exists (read it), 8 = issue the DMA of tile t+2.
  // sr = Sᵀ(j) (read), sw = Sᵀ(j+1) (written), pw = P(j) (written), pr = P(j−1) (read).
  auto phase = [&](auto hc, auto fc, int t, f32x4_t (&sr)[2][4], f32x4_t (&sw)[2][4], half8_t (&pw)[4], half8_t (&pr)[4]) {
    constexpr int H = decltype(hc)::value, F = decltype(fc)::value;
    constexpr bool HAS_PV = (F & 1) != 0, HAS_QK = (F & 2) != 0, HAS_KRD = (F & 4) != 0, HAS_DMA = (F & 8) != 0;
    constexpr int KQ = AN_K + 32 * (1 - H);   // K(j+1) fragments: AGPR buffer (j+1) & 1
    constexpr int KRB = H;                     // K(j+2) goes to buffer (j+2) & 1 = H
    uint32_t (&vb_a)[4] = H == 0 ? vp : vc;    // Vᵀ(j−1) lives in tile t−1 (H = 0, its second half) or tile t (H = 1, first half)
    constexpr int VB_H = H == 0 ? 1 : 0;       // half-tile of Vᵀ(j−1) inside its tile
    // set A (db 0..3, read in slots 32..39 of the previous phase) and K(j+1) are needed from slot 0 / 1 on
    am_wait_v8(reinterpret_cast<half4_t(&)[4]>(vlo[0]), reinterpret_cast<half4_t(&)[4]>(vhi[0]));
    float ps[4][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    float e0 = 0.f, e1 = 0.f, c0 = 0.f, c1 = 0.f;   // exps of the pair in flight / of the pair being packed
    // a pair (two consecutive kv of one query row) is finished in two halves, each placed IN FRONT of a v_exp of the
    // next pair: hipcc pads a wait state between an asm statement and a transcendental that follows it directly
    auto pair_sum = [&](auto pc, auto wc, float a) {     // row sums from the unrounded P (split_q.cu:467-468)
      constexpr int qb = (decltype(pc)::value >> 1) & 3, w = decltype(wc)::value;
      ps[qb][w] += a;
      asm volatile("" : "+v"(ps[qb][w]));
    };
    auto pair_pack = [&](auto pc, float a, float b) {
      constexpr int p = decltype(pc)::value, kvb = p >> 3, qb = (p >> 1) & 3, k2 = p & 1;
      half2_t h2 = {(half_t)a, (half_t)b};
      asm volatile("" : "+v"(h2));
      pw[qb][4 * kvb + 2 * k2] = h2[0];
      pw[qb][4 * kvb + 2 * k2 + 1] = h2[1];
    };
    static_for<64>([&](auto sc) {
      constexpr int s = decltype(sc)::value, i = s >> 1;
      // ---------------- the MFMA of this slot + the LDS reads in its shadow (one asm statement)
      // slots 0..7: Vᵀ(j−1) set B, transpose read c = s (db = 4 + (c >> 1), kv block c & 1), and K(j+2) fragment c = s;
      // slots 32..39: Vᵀ(j) set A, transpose read c = s − 32 (db = c >> 1)
      constexpr bool RVB = s < 8 && HAS_PV, RK = s < 8 && HAS_KRD, RVA = s >= 32 && s < 40;
      constexpr int RD = ((RVB || RVA) ? 1 : 0) | (RK ? 2 : 0);
      constexpr int c = RVA ? s - 32 : (s & 7), rdb = (RVA ? 0 : 4) + (c >> 1), rx = c & 1;
      constexpr int VOF = (RVA ? H : VB_H) * 8192 + rx * 4096;
      half4_t& vout = rx ? vhi[rdb] : vlo[rdb];
      const uint32_t vaddr = RVA ? vaddr_of(vc, rdb) : vaddr_of(vb_a, rdb);
      constexpr int KR = AN_K + 32 * KRB + 4 * (s & 7), KOF = H * 8192 + ((s & 7) >> 2) * 4096;
      if constexpr ((s & 1) == 0) {
        constexpr int ds = i >> 3, kvb = (i >> 2) & 1, qb = i & 3;
        constexpr int KIND = HAS_QK ? (ds == 0 ? 0 : 1) : 3;
        if constexpr (KIND != 3 || RD != 0)
          an_slot<KIND, RD, KQ + 4 * (4 * kvb + ds), AN_Q + 4 * (4 * qb + ds), VOF, KR, KOF>(
              sw[kvb][qb], negm[qb], half8_t{}, half8_t{}, vout, vaddr, ka[s & 3]);
      } else {
        constexpr int db = i >> 2, qb = i & 3;
        constexpr int KIND = HAS_PV ? 2 : 3;
        if constexpr (KIND != 3 || RD != 0)
          an_slot<KIND, RD, AN_O + 4 * (4 * db + qb), 0, VOF, KR, KOF>(sw[0][0], negm[0], cat4(vlo[db], vhi[db]), pr[qb], vout,
                                                                       vaddr, ka[s & 3]);
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- compiler-scheduled fillers behind it
      if constexpr (s == 31 && HAS_PV)   // set B was read in slots 0..7; its first P·V MFMA is slot 33
        am_wait_v8(reinterpret_cast<half4_t(&)[4]>(vlo[4]), reinterpret_cast<half4_t(&)[4]>(vhi[4]));
      if constexpr (HAS_DMA && (s & 7) == 7) issue_piece(s >> 3, t + 2);                  // 8 pieces, one per 8 slots
      // softmax(j): pair p = s >> 2 -> kv block p >> 3, query block (p >> 1) & 3, values 2 k2, 2 k2 + 1 of the 4.
      // slot 4p: first row sum of pair p − 1, then v_exp of pair p's first value; slot 4p + 1: second row sum; slot 4p + 2: fp16
      // pack of pair p − 1, then v_exp of the second value
      if constexpr ((s & 3) == 0) {
        constexpr int p = s >> 2, kvb = p >> 3, qb = (p >> 1) & 3, k2 = p & 1;
        if constexpr (p >= 1) {
          pair_sum(std::integral_constant<int, p - 1>{}, I0{}, e0);
          c0 = e0;
          c1 = e1;
          __builtin_amdgcn_sched_barrier(0);
        }
        e0 = __builtin_amdgcn_exp2f(sr[kvb][qb][2 * k2]);
        asm volatile("" : "+v"(e0));
      } else if constexpr ((s & 3) == 1) {   // (a plain VALU between two asm statements also saves hipcc's boundary s_nop)
        if constexpr (s >= 5) pair_sum(std::integral_constant<int, (s >> 2) - 1>{}, I1{}, c1);
      } else if constexpr ((s & 3) == 2) {
        constexpr int p = s >> 2, kvb = p >> 3, qb = (p >> 1) & 3, k2 = p & 1;
        if constexpr (p >= 1) {
          pair_pack(std::integral_constant<int, p - 1>{}, c0, c1);
          __builtin_amdgcn_sched_barrier(0);
        }
        e1 = __builtin_amdgcn_exp2f(sr[kvb][qb][2 * k2 + 1]);
        asm volatile("" : "+v"(e1));
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    pair_sum(std::integral_constant<int, 15>{}, I0{}, e0);
    pair_sum(std::integral_constant<int, 15>{}, I1{}, e1);
    pair_pack(std::integral_constant<int, 15>{}, e0, e1);
    // ---------------- overflow guard: m is only a scale; redo this half-tile with the true max when P got large
    // (bit patterns of non-negative floats order like unsigned integers, NaN / inf sit above every finite limit: ONE integer
    // compare of the largest pattern decides for the four query blocks — lc_common.h psum_below)
    uint32_t worst_bits = 0;
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) worst_bits = max(worst_bits, __builtin_bit_cast(uint32_t, ps[qb][0] + ps[qb][1]));
    const bool ok = worst_bits < __builtin_bit_cast(uint32_t, AM_PSUM_LIMIT);
    if (!__all(ok)) {                                          // NaN / inf take this path too
      am_drain(sw);                                            // every MFMA of this phase has written its result
      {
        float worst = 0.f;
        bool fin = true;
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
          const float x = ps[qb][0] + ps[qb][1];
          fin = fin && finite_bits(x);
          if (!psum_below(x, AM_PSUM_LIMIT)) worst = x;
        }
        const unsigned long long culprit = __ballot(!ok);
        if (lane == (int)__builtin_ctzll(culprit | (1ull << 63))) {
          atomicAdd(&LC_AN_SLOWPATH_SYM[0], 1u);
          atomicAdd(&LC_AN_SLOWPATH_SYM[1], (unsigned)(2 * t + H));
          if (!fin) atomicAdd(&LC_AN_SLOWPATH_SYM[2], 1u);
          LC_AN_SLOWPATH_SYM[3] = __builtin_bit_cast(unsigned, worst);
        }
      }
      static_for<4>([&](auto qc) {
        constexpr int qb = decltype(qc)::value;
        float mx = sr[0][qb][0];
#pragma unroll
        for (int kvb = 0; kvb < 2; ++kvb)
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sr[kvb][qb][r]);
        mx = an_x4_max(mx);
        const float delta = fmaxf(mx, 0.f);                    // the row's max grew by `delta` (log2 units)
        const float alpha = __builtin_amdgcn_exp2f(-delta);
        l_run[qb] *= alpha;
        ps[qb][0] = 0.f;
        ps[qb][1] = 0.f;
#pragma unroll
        for (int kvb = 0; kvb < 2; ++kvb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if constexpr (HAS_QK) sw[kvb][qb][r] -= delta;       // Sᵀ(j+1) was accumulated against the old max
            const float pv = __builtin_amdgcn_exp2f(sr[kvb][qb][r] - delta);
            ps[qb][r & 1] += pv;
            pw[qb][4 * kvb + r] = (half_t)pv;
          }
#pragma unroll
        for (int r = 0; r < 4; ++r) negm[qb][r] -= delta;
        static_for<8>([&](auto dc) {
          constexpr int db = decltype(dc)::value;
          static_for<4>([&](auto rc) { am_acc_scale<AN_O + 4 * (4 * db + qb) + decltype(rc)::value>(alpha); });
        });
      });
      asm volatile("s_nop 3" ::: "memory");    // VALU writes of −m / S / P -> MFMA operand reads of the next phase
    }
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) l_run[qb] += ps[qb][0] + ps[qb][1];
  };
  using F_FIRST0 = std::integral_constant<int, 2 | 4 | 8>;       // j = 0: no P·V(−1)
  using F_MID = std::integral_constant<int, 1 | 2 | 4 | 8>;
  using F_MID1 = std::integral_constant<int, 1 | 2 | 4>;         // odd phases do not issue DMA
  using F_LAST0 = std::integral_constant<int, 1 | 2>;            // j = 2T−2: no tile T to read K from / to stage
  using F_LAST1 = std::integral_constant<int, 1>;                // j = 2T−1: no Q·Kᵀ(2T)

  // Vᵀ(0) set A (db 0..3) for the P·V of phase 1: read here, retired by phase 1's opening wait... phase 0 reads it in its
  // slots 32..39 like every phase (RVA) — nothing to do in the prologue.
  set_tile_addrs(0);
  phase(I0{}, F_FIRST0{}, 0, sA, sB, pA, pB);
  phase(I1{}, F_MID1{}, 0, sB, sA, pB, pA);
  for (int t = 1; t + 1 < T; ++t) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // own pieces of tile t+1 landed; own LDS reads retired
      raw_barrier();
    set_tile_addrs(t);
    phase(I0{}, F_MID{}, t, sA, sB, pA, pB);
    phase(I1{}, F_MID1{}, t, sB, sA, pB, pA);
  }
  {
    const int t = T - 1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      raw_barrier();
    set_tile_addrs(t);
    phase(I0{}, F_LAST0{}, t, sA, sB, pA, pB);
    phase(I1{}, F_LAST1{}, t, sB, sA, pB, pA);
    // tail: Oᵀ += Vᵀ(2T−1)·Pᵀ(2T−1); set A was read in the last phase, set B (second half of the last tile) now
    static_for<8>([&](auto cc) {
      constexpr int c = decltype(cc)::value, db = 4 + (c >> 1);
      if constexpr ((c & 1) == 0) vlo[db] = lds_tr16_asm<8192>(vaddr_of(vc, db));
      else vhi[db] = lds_tr16_asm<8192 + 4096>(vaddr_of(vc, db));
    });
    am_wait_v8(reinterpret_cast<half4_t(&)[4]>(vlo[0]), reinterpret_cast<half4_t(&)[4]>(vhi[0]));
    am_wait_v8(reinterpret_cast<half4_t(&)[4]>(vlo[4]), reinterpret_cast<half4_t(&)[4]>(vhi[4]));
    static_for<32>([&](auto ic) {
      constexpr int i = decltype(ic)::value, db = i >> 2, qb = i & 3;
      an_pv<AN_O + 4 * (4 * db + qb)>(cat4(vlo[db], vhi[db]), pB[qb]);
    });
  }

  // ---- epilogue: O = Oᵀ / l through LDS (whole 256-B rows, 16-B stores).  Lane holds O[q = 16 qb + l16][d = 16 db + 4 g4 +
  // (0..3)] in a[4 (4 db + qb) ..]; every wave owns a private 64 x 272 B staging area.
  am_drain();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  raw_barrier();                     // every wave is done with the KV ring
  float inv[4];
#pragma unroll
  for (int qb = 0; qb < 4; ++qb) inv[qb] = 1.0f / an_x4_sum(l_run[qb]);
  char* stg = smem + wave * (64 * AM_EPI_STRIDE);
  static_for<4>([&](auto qc) {
    constexpr int qb = decltype(qc)::value;
    static_for<8>([&](auto dc) {
      constexpr int db = decltype(dc)::value;
      constexpr int base = AN_O + 4 * (4 * db + qb);
      half4_t h;
      h[0] = (half_t)(am_acc_read<base + 0>() * inv[qb]);
      h[1] = (half_t)(am_acc_read<base + 1>() * inv[qb]);
      h[2] = (half_t)(am_acc_read<base + 2>() * inv[qb]);
      h[3] = (half_t)(am_acc_read<base + 3>() * inv[qb]);
      *(half4_t*)(stg + (16 * qb + l16) * AM_EPI_STRIDE + (16 * db + 4 * g4) * 2) = h;
    });
  });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private staging: own writes visible to own reads
  half_t* ow = Ob + (size_t)q0 * D;
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int row = it * 4 + (lane >> 4);
    const u32x4_t v = *(const u32x4_t*)(stg + row * AM_EPI_STRIDE + (lane & 15) * 16);
    *(u32x4_t*)(ow + (size_t)row * D + (lane & 15) * 8) = v;
  }
}

}  // namespace lc
