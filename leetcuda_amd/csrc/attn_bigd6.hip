// attn_bigd6.hip — FlashAttention-2 forward, D = 512 (fp16 / bf16): attn_bigd2's pipeline on v_mfma_f32_16x16x32 (round 4).
//
// Reference: kernels/flash-attn/mma/basic/flash_attn_mma_tiling_qkv.cu:75-797 (entry :881-945); BASELINE config 5a (1,48,8192,512).
//
// Why: config 5a runs at the board's 1400 W cap (attn_bigd2: 1.20 PFLOP/s on randn, 1.52 zero-filled), and at the cap an MFMA-only
// stream of v_mfma_f32_16x16x32_f16 sustains 14 % more FLOP/s than one of 32x32x16 (1854 vs 1625 TFLOP/s, DESIGN.md §4.10: half the
// accumulator registers moved per FLOP).  attn_bigd2 sits at 0.74 of ITS form's ceiling; with the rest of its energy per FLOP unchanged
// the 16-wide form projects to ≈ 1.32 PFLOP/s.  Measured (profiles/r4k_bigd6.log, same box, interleaved): fp16 1195 - 1205 vs 1142 - 1165
// TFLOP/s (+ 3.4 ... 4.7 %), bf16 1228 - 1265 vs 1206 - 1231 (+ 1.8 ... 2.8 %); zero-filled 1382 vs 1508 — twice the MFMA statements put
// the one-wave-per-SIMD instruction stream closer to its issue limit (≈ 720 instructions per tile period of 4096 matrix-core
// cycles), which is what the cap hides.  Default for D = 512 since; attn_bigd2 stays for D = 256, V-transposed inputs and as the
// cross-check on the other MFMA shape (lc_tune_set "attn_d512" = 3).  Same work split (four wave64, a wave = 32 query rows x all 512 columns, Oᵀ in the 256
// AGPRs, Q fragments loaded once, K / V tiles of 64 rows single-buffered in 2 x 64 KiB and filled by LDS-DMA in the shadow of the other
// phase), same K image, every fragment layout re-derived for the 16-wide shapes (they are attn_w4u's, on 1-KiB rows):
//   * wave = 2 query blocks qb of 16 rows; KV tile = 4 kv blocks kvb of 16;
//   * Sᵀ block (kvb, qb) = Σ_ds K fragment (kvb, ds) x Q fragment (qb, ds), ds = 0 .. 15 (32 d each): lane (l16, g4) holds
//     S[q = 16 qb + l16][kv = 16 kvb + 4 g4 + r] — 8 blocks x 4 registers, each touched by every 8th MFMA;
//   * the Pᵀ operand is lane-local: P(qb, h) = pack(S(2h, qb)[0..3], S(2h + 1, qb)[0..3]), k slot 8 g4 + e <-> kv = 32 h + 16 (e >> 2) +
//     4 g4 + (e & 3); the Vᵀ operand follows with two ds_read_b64_tr_b16 per fragment (kv rows 4 g4 .. of kv block 2h, then 2h + 1);
//   * Oᵀ block (db, qb) = a[4 (2 db + qb) ..]: lane holds O[q = 16 qb + l16][d = 16 db + 4 g4 + r];
//   * V tile: 32-B column pair p of row r at pair slot (p & ~7) | ((p ^ key(r)) & 7), key = ((r & 3) << 1) | ((r >> 2) & 1)
//     (attn_w4u's D = 128 key on 1-KiB rows: conflict-free for the 2 x 32 lane groups of the transpose reads, tests/test_layouts.py);
//   * a P·V step = four Vᵀ fragments in the fixed quads v[240:255] x both query blocks = 8 MFMAs in ONE statement, each fragment's
//     two transpose reads for the NEXT step issued right behind its two MFMAs (attn_bigd2's counted lgkmcnt(6) protocol).
#pragma once
#include "attn_bigd2.hip"

namespace lc {

// the eight MFMAs of one d-step — Sᵀ blocks (kvb, qb) += K fragment (kvb) x Q fragment (qb), kvb = 0 .. 3, qb = 0, 1 — in ONE statement
// (hipcc pads every asm boundary with a wait state: eight one-MFMA statements per 128 matrix-core cycles cost 7 of them).  FIRST: the
// first d-step of a tile takes the inline constant 0 as its accumulator input (write-only outputs: no VALU zeroing of the 32 score
// registers — attn_bigd7's trick, 16 v_mov_b64 per tile less)
template <bool BF16, bool FIRST>
LC_DEVINL void bd6_qk8(f32x4_t (&s)[4][2], const half8_t (&k)[4], half8_t q0, half8_t q1) {
#define LC_BD6_QK8(OP)                                                                                                            \
  asm volatile(OP " %0, %8, %12, %0\n\t" OP " %1, %8, %13, %1\n\t" OP " %2, %9, %12, %2\n\t" OP " %3, %9, %13, %3\n\t"                   \
               OP " %4, %10, %12, %4\n\t" OP " %5, %10, %13, %5\n\t" OP " %6, %11, %12, %6\n\t" OP " %7, %11, %13, %7"                  \
               : "+v"(s[0][0]), "+v"(s[0][1]), "+v"(s[1][0]), "+v"(s[1][1]), "+v"(s[2][0]), "+v"(s[2][1]), "+v"(s[3][0]), "+v"(s[3][1])  \
               : "v"(k[0]), "v"(k[1]), "v"(k[2]), "v"(k[3]), "v"(q0), "v"(q1)                                                        \
               : LC_AGPR_ALL)
#define LC_BD6_QK8_FIRST(OP)                                                                                                      \
  asm volatile(OP " %0, %8, %12, 0\n\t" OP " %1, %8, %13, 0\n\t" OP " %2, %9, %12, 0\n\t" OP " %3, %9, %13, 0\n\t"                       \
               OP " %4, %10, %12, 0\n\t" OP " %5, %10, %13, 0\n\t" OP " %6, %11, %12, 0\n\t" OP " %7, %11, %13, 0"                      \
               : "=&v"(s[0][0]), "=&v"(s[0][1]), "=&v"(s[1][0]), "=&v"(s[1][1]), "=&v"(s[2][0]), "=&v"(s[2][1]), "=&v"(s[3][0]),        \
                 "=&v"(s[3][1])                                                                                                    \
               : "v"(k[0]), "v"(k[1]), "v"(k[2]), "v"(k[3]), "v"(q0), "v"(q1)                                                        \
               : LC_AGPR_ALL)
  if constexpr (FIRST) {
    if constexpr (BF16) { LC_BD6_QK8_FIRST("v_mfma_f32_16x16x32_bf16"); }
    else { LC_BD6_QK8_FIRST("v_mfma_f32_16x16x32_f16"); }
  } else {
    if constexpr (BF16) { LC_BD6_QK8("v_mfma_f32_16x16x32_bf16"); }
    else { LC_BD6_QK8("v_mfma_f32_16x16x32_f16"); }
  }
#undef LC_BD6_QK8_FIRST
#undef LC_BD6_QK8
}
// P·V step: Oᵀ blocks (db = 4 s + j, qb) += Vᵀ fragment j (fixed quad) x Pᵀ(qb), j = 0 .. 3; RD: + the two transpose reads of the NEXT
// step's fragment j into the same quad (address A_j, offsets OFF / OFF + HOFF).  R0 = 32 s: block (db, qb) = a[R0 + 8 j + 4 qb ..].
// Entry: 8 reads outstanding, in fragment order.  The leading s_nop 1: hipcc packs a P fragment right in front of the statement.
template <int R0, bool BF16, bool RD, int OFF, int HOFF>
LC_DEVINL void bd6_pv8_fix(half8_t& f0, half8_t& f1, half8_t& f2, half8_t& f3, half8_t p0, half8_t p1, uint32_t a0, uint32_t a1,
                           uint32_t a2, uint32_t a3) {
#define LC_BD6_STEP(OP, W0, W1, W2, W3, R0_, R1_, R2_, R3_)                                                                                  \
  asm volatile("s_nop 1\n\ts_waitcnt lgkmcnt(" #W0 ")\n\t" OP " a[%10:%11], v[240:243], %4, a[%10:%11]\n\t" OP " a[%12:%13], v[240:243], %5, a[%12:%13]\n\t" R0_ \
               "s_waitcnt lgkmcnt(" #W1 ")\n\t" OP " a[%14:%15], v[244:247], %4, a[%14:%15]\n\t" OP " a[%16:%17], v[244:247], %5, a[%16:%17]\n\t" R1_            \
               "s_waitcnt lgkmcnt(" #W2 ")\n\t" OP " a[%18:%19], v[248:251], %4, a[%18:%19]\n\t" OP " a[%20:%21], v[248:251], %5, a[%20:%21]\n\t" R2_            \
               "s_waitcnt lgkmcnt(" #W3 ")\n\t" OP " a[%22:%23], v[252:255], %4, a[%22:%23]\n\t" OP " a[%24:%25], v[252:255], %5, a[%24:%25]\n\t" R3_            \
               : "+{v[240:243]}"(f0), "+{v[244:247]}"(f1), "+{v[248:251]}"(f2), "+{v[252:255]}"(f3)                                          \
               : "v"(p0), "v"(p1), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "n"(R0), "n"(R0 + 3), "n"(R0 + 4), "n"(R0 + 7), "n"(R0 + 8),            \
                 "n"(R0 + 11), "n"(R0 + 12), "n"(R0 + 15), "n"(R0 + 16), "n"(R0 + 19), "n"(R0 + 20), "n"(R0 + 23), "n"(R0 + 24), "n"(R0 + 27),  \
                 "n"(R0 + 28), "n"(R0 + 31), "n"(OFF), "n"(OFF + HOFF)                                                                        \
               : LC_AGPR_ALL)
#define LC_BD6_RDS(OP)                                                                                                                       \
  LC_BD6_STEP(OP, 6, 6, 6, 6, "ds_read_b64_tr_b16 v[240:241], %6 offset:%26\n\tds_read_b64_tr_b16 v[242:243], %6 offset:%27\n\t",             \
              "ds_read_b64_tr_b16 v[244:245], %7 offset:%26\n\tds_read_b64_tr_b16 v[246:247], %7 offset:%27\n\t",                               \
              "ds_read_b64_tr_b16 v[248:249], %8 offset:%26\n\tds_read_b64_tr_b16 v[250:251], %8 offset:%27\n\t",                               \
              "ds_read_b64_tr_b16 v[252:253], %9 offset:%26\n\tds_read_b64_tr_b16 v[254:255], %9 offset:%27")
  if constexpr (RD) {
    if constexpr (BF16) LC_BD6_RDS("v_mfma_f32_16x16x32_bf16");
    else LC_BD6_RDS("v_mfma_f32_16x16x32_f16");
  } else {
    if constexpr (BF16) LC_BD6_STEP("v_mfma_f32_16x16x32_bf16", 6, 4, 2, 0, "", "", "", "");
    else LC_BD6_STEP("v_mfma_f32_16x16x32_f16", 6, 4, 2, 0, "", "", "", "");
  }
#undef LC_BD6_RDS
#undef LC_BD6_STEP
}
// the eight transpose reads of a step's four fragments (kv rows at OFF, second kv block at OFF + HOFF), in fragment order
template <int OFF, int HOFF>
LC_DEVINL void bd6_rd(half8_t& f0, half8_t& f1, half8_t& f2, half8_t& f3, uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3) {
  asm volatile("ds_read_b64_tr_b16 v[240:241], %4 offset:%8\n\tds_read_b64_tr_b16 v[242:243], %4 offset:%9\n\t"
               "ds_read_b64_tr_b16 v[244:245], %5 offset:%8\n\tds_read_b64_tr_b16 v[246:247], %5 offset:%9\n\t"
               "ds_read_b64_tr_b16 v[248:249], %6 offset:%8\n\tds_read_b64_tr_b16 v[250:251], %6 offset:%9\n\t"
               "ds_read_b64_tr_b16 v[252:253], %7 offset:%8\n\tds_read_b64_tr_b16 v[254:255], %7 offset:%9"
               : "={v[240:243]}"(f0), "={v[244:247]}"(f1), "={v[248:251]}"(f2), "={v[252:255]}"(f3)
               : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "n"(OFF), "n"(OFF + HOFF));
}

template <bool BF16>
__global__ __launch_bounds__(256) void attn_fwd_bigd6_kernel(
    const half_t* __restrict__ Q, const half_t* __restrict__ K, const half_t* __restrict__ V,
    half_t* __restrict__ O, int N, int nqb_arg, float sl2) {
  constexpr int D = 512;
  constexpr int ROWB = D * 2;              // bytes per K / V row (1 KiB)
  constexpr int TILE = KVB * ROWB;         // one K or V tile (64 KiB)
  constexpr int NDS = D / 32;              // d-steps of Q·Kᵀ (16)
  constexpr int NDB = D / 16;              // 16-column Oᵀ blocks (32)
  constexpr int NPIECE = TILE / 1024 / 4;  // DMA pieces per wave and tile (16): a piece = one row
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = wave_id();
  const int g4 = lane >> 4, l16 = lane & 15;

  // nqb_arg < 0 (lc_tune_set "attn_bigd_map"): block b is query block b of the launch — consecutive blocks of a head go round-robin
  // over the 8 XCDs, so EVERY XCD streams the head's K / V for its share of the blocks: about twice the fabric bytes of the XCD-contiguous
  // map (nqb_arg > 0: one XCD owns consecutive blocks, bench.py attn_traffic_model), but all XCDs walk the same two heads, whose K / V
  // then live in the Infinity Cache.  Measured (profiles/r5f_bigd_map.log, r5f_bigd_map_pmc.log): D = 1024 + 3.7 % at 13.7 vs 7.2 GB
  // fetched, D = 512 - 2 % at 6.8 vs 2.0 GB — fabric bytes are not what bounds these kernels.  Same bits either way.
  const int nqb = nqb_arg < 0 ? -nqb_arg : nqb_arg;
  const int id = __builtin_amdgcn_readfirstlane(nqb_arg < 0 ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x));
  const int bhi = __builtin_amdgcn_readfirstlane(id / nqb);
  const size_t bh = (size_t)bhi;
  const int q0 = __builtin_amdgcn_readfirstlane((id - bhi * nqb) * 128 + wave * 32);
  const half_t* Qb = Q + bh * (size_t)N * D;
  const char* Kb = (const char*)(K + bh * (size_t)N * D);
  const char* Vb = (const char*)(V + bh * (size_t)N * D);
  half_t* Ob = O + bh * (size_t)N * D;
  const int T = N / KVB;
  const uint32_t smem32 = lds_addr32(smem);
  char* const ksm = smem;
  char* const vsm = smem + TILE;

  // ---- LDS-DMA: piece = one 1-KiB row; this wave stages rows wave + 4 i.  Lane chunk slot cs holds source chunk cs ^ key(row)
  // (K: row & 15 in the low 4 bits -> k_off[i & 3]; V: pair (cs >> 1) ^ key(row) in the low 3 bits, key(row) = ((row & 3) << 1) |
  // ((row >> 2) & 1) = (wave << 1) | (i & 1) -> v_off[i & 1])
  const buf_rsrc_t rk = make_rsrc(Kb), rv = make_rsrc(Vb);
  unsigned k_off[4], v_off[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) k_off[j] = (unsigned)((lane ^ ((wave + 4 * j) & 15)) * 16);
#pragma unroll
  for (int j = 0; j < 2; ++j) v_off[j] = (unsigned)((((((lane >> 1) ^ ((wave << 1) | j)) & 7) | ((lane >> 1) & ~7)) << 1 | (lane & 1)) * 16);
  auto issue_k = [&](int i, int t) {
    const int te = t < T ? t : T - 1;
    const int p = wave + 4 * i;
    blds16(rk, k_off[i & 3], (unsigned)te * TILE + (unsigned)p * 1024u, ksm + p * 1024);
  };
  auto issue_v = [&](int i, int t) {
    const int te = t < T ? t : T - 1;
    const int p = wave + 4 * i;
    blds16(rv, v_off[i & 1], (unsigned)te * TILE + (unsigned)p * 1024u, vsm + p * 1024);
  };
#pragma unroll
  for (int i = 0; i < NPIECE; ++i) issue_k(i, 0);

  // ---- Q fragments -> registers (once): lane holds Q[q0 + 16 qb + l16][32 ds + 8 g4 .. +8]; the last PARK d-steps are parked in
  // the 32 KiB of LDS the tiles leave free (lane-private 16-B slots) and come back through a register ring during Q·Kᵀ
  constexpr int PARK = 4, NRES = NDS - PARK;
  half8_t qf[NRES][2];
#pragma unroll
  for (int ds = 0; ds < NRES; ++ds)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) qf[ds][qb] = *(const half8_t*)(Qb + (size_t)(q0 + 16 * qb + l16) * D + 32 * ds + 8 * g4);
  char* const qpark = smem + 2 * TILE + wave * (PARK * 2048) + lane * 16;   // + 2048 per parked d-step, + 1024 for qb = 1
#pragma unroll
  for (int i = 0; i < PARK; ++i)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
      *(half8_t*)(qpark + i * 2048 + qb * 1024) = *(const half8_t*)(Qb + (size_t)(q0 + 16 * qb + l16) * D + 32 * (NRES + i) + 8 * g4);
  static_for<D / 2>([&](auto r) { am_acc_zero<decltype(r)::value>(); });

  // ---- fragment read addresses
  const char* kx[4];   // K: row l16 (+ 16 kvb), chunk 4 ds + g4: low 4 bits XOR (row & 15) = l16; + (ds >> 2) * 256 as immediate
#pragma unroll
  for (int k4 = 0; k4 < 4; ++k4) kx[k4] = ksm + l16 * ROWB + (((4 * k4 + g4) ^ l16) * 16);
  // Vᵀ: kv row 4 g4 + (l16 >> 2) (+ 16 x, + 32 h: immediates), 8 bytes at column 4 (l16 & 3) of pair db: slot ((db & 7) ^ key) + 8 (db >> 3),
  // key = ((l16 >> 2) << 1) | (g4 & 1) -> vx[db & 7], + (db >> 3) * 256 as immediate
  uint32_t vx[8];
#pragma unroll
  for (int b = 0; b < 8; ++b)
    vx[b] = smem32 + (uint32_t)(TILE + (4 * g4 + (l16 >> 2)) * ROWB + 8 * (l16 & 3) + ((b ^ (((l16 >> 2) << 1) | (g4 & 1))) * 32));

  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
  half8_t pfa[2][2], pfb[2][2];   // P fragments [qb][h] of the even / odd tiles

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  raw_barrier();   // K(0) landed

  // ---- P·V step st = (h, s): kv half-tile h (32 rows), column blocks db = 4 s .. 4 s + 3, both query blocks
  half8_t vf0, vf1, vf2, vf3;
  constexpr int NSB = NDB / 4, NST = 2 * NSB;   // 8 steps per half-tile, 16 per tile
  constexpr int SP8 = 6;                        // DMA pieces spread over 6/8 of a phase (attn_bigd2.hip: TA FIFO)
  constexpr int SPAN_A = SP8 * NDS / 8, SPAN_B = SP8 * NST / 8;
  auto rd0 = [&]() { bd6_rd<0, 16 * ROWB>(vf0, vf1, vf2, vf3, vx[0], vx[1], vx[2], vx[3]); };
  auto pv_step = [&](auto stc, half8_t (&pf)[2][2]) {
    constexpr int st = decltype(stc)::value, h = st / NSB, s = st % NSB;
    constexpr int h1 = (st + 1) / NSB, s1 = (st + 1) % NSB, o1 = 4 * (s1 & 1);
    bd6_pv8_fix<32 * s, BF16, (st + 1 < NST), (s1 >> 1) * 256 + h1 * 32 * ROWB, 16 * ROWB>(
        vf0, vf1, vf2, vf3, pf[0][h], pf[1][h], vx[o1], vx[o1 + 1], vx[o1 + 2], vx[o1 + 3]);
  };

  // ---- one tile period (attn_bigd2.hip's): phase A = Sᵀ(t) = K(t)·Qᵀ with the DMA of V(t−1); barrier; phase B = P·V(t−1) with
  // softmax(t) as filler and the DMA of K(t+1); barrier.  pn = P(t) (written), po = P(t−1) (read).  HAS_PV = false: tile 0.
  auto tile = [&](auto pvc, int t, half8_t (&pn)[2][2], half8_t (&po)[2][2]) {
    constexpr bool HAS_PV = decltype(pvc)::value;
    f32x4_t s[4][2];   // [kvb][qb]; written by the first d-step (accumulator input 0)
    {
      half8_t kfr[2][4], qfr[2][2];   // K fragments / parked Q fragments of d-step ds in ring slot ds & 1
      auto ldk = [&](auto dc) {
        constexpr int ds = decltype(dc)::value, r = ds & 1;
        static_for<4>([&](auto kc) {
          constexpr int kvb = decltype(kc)::value;
          kfr[r][kvb] = *(const half8_t*)(kx[ds & 3] + (ds >> 2) * 256 + kvb * 16 * ROWB);
        });
        if constexpr (ds >= NRES) {
          qfr[r][0] = *(const half8_t*)(qpark + (ds - NRES) * 2048);
          qfr[r][1] = *(const half8_t*)(qpark + (ds - NRES) * 2048 + 1024);
        }
      };
      ldk(std::integral_constant<int, 0>{});
      static_for<NDS>([&](auto dc) {
        constexpr int ds = decltype(dc)::value;
        if constexpr (ds + 1 < NDS) ldk(std::integral_constant<int, ds + 1>{});
        if constexpr (HAS_PV)
          static_for<NPIECE>([&](auto ic) {
            if constexpr (decltype(ic)::value * SPAN_A / NPIECE == ds) issue_v(decltype(ic)::value, t - 1);
          });
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ds < NRES) bd6_qk8<BF16, ds == 0>(s, kfr[ds & 1], qf[ds][0], qf[ds][1]);
        else bd6_qk8<BF16, false>(s, kfr[ds & 1], qfr[ds & 1][0], qfr[ds & 1][1]);
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    // asm MFMAs: hipcc does not know their latency; VALU reads S next (the registers are operands of the drain: rule R5)
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15"
                 : "+v"(s[0][0]), "+v"(s[0][1]), "+v"(s[1][0]), "+v"(s[1][1]), "+v"(s[2][0]), "+v"(s[2][1]), "+v"(s[3][0]), "+v"(s[3][1])
                 :: "memory");
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // own V(t−1) pieces landed, own K reads retired
    raw_barrier();                                                // K(t) is dead, V(t−1) complete

    // =========================== phase B
    float ps[2] = {0.f, 0.f};
    const float nm[2] = {-m_run[0], -m_run[1]};
    if constexpr (HAS_PV) rd0();
    static_for<NST>([&](auto stc) {
      constexpr int st = decltype(stc)::value;
      static_for<NPIECE>([&](auto ic) {
        if constexpr (decltype(ic)::value * SPAN_B / NPIECE == st) issue_k(decltype(ic)::value, t + 1);
      });
      if constexpr (HAS_PV) pv_step(stc, po);
      __builtin_amdgcn_sched_barrier(0);
      // softmax(t): two score elements per step (32 per lane and tile), row sums from the unrounded P (tiling_qkv.cu's order).
      // element e = 2 st + j -> block (kvb, qb) = (e >> 3, (e >> 2) & 1), register e & 3; P fragment (qb, h = kvb >> 1), slot 4 (kvb & 1) + r
      static_for<2>([&](auto jc) {
        constexpr int e = 2 * st + decltype(jc)::value, kvb = e >> 3, qb = (e >> 2) & 1, r = e & 3;
        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kvb][qb][r], sl2, nm[qb]));
        ps[qb] += p;
        pn[qb][kvb >> 1][4 * (kvb & 1) + r] = cvt16<BF16>(p);
      });
      __builtin_amdgcn_sched_barrier(0);
    });
    if (!__all(psum_below(ps[0], 16384.0f) && psum_below(ps[1], 16384.0f)) || !HAS_PV) {   // overflow guard / first tile: the true max
      am_drain();    // the P·V MFMAs of this phase have written Oᵀ
      static_for<2>([&](auto qc) {
        constexpr int qb = decltype(qc)::value;
        float mx = s[0][qb][0];
#pragma unroll
        for (int kvb = 0; kvb < 4; ++kvb)
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kvb][qb][r]);
        mx = an_x4_max(mx * sl2);                      // (sl2 > 0); a row's kv columns are spread over the four 16-lane groups
        const float m_new = fmaxf(m_run[qb], mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_new);   // exp2(-inf) = 0 on the first tile
        m_run[qb] = m_new;
        l_run[qb] *= alpha;
        static_for<NDB>([&](auto dc) {
          static_for<4>([&](auto rc) { am_acc_scale<4 * (2 * decltype(dc)::value + qb) + decltype(rc)::value>(alpha); });
        });
        ps[qb] = 0.f;
#pragma unroll
        for (int kvb = 0; kvb < 4; ++kvb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kvb][qb][r], sl2, -m_run[qb]));
            ps[qb] += p;
            pn[qb][kvb >> 1][4 * (kvb & 1) + r] = cvt16<BF16>(p);
          }
      });
    }
    l_run[0] += ps[0];
    l_run[1] += ps[1];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // own K(t+1) pieces landed, own V reads retired
    raw_barrier();                                                // V(t−1) is dead, K(t+1) complete
  };
  using HAS = std::integral_constant<bool, true>;
  using HASNOT = std::integral_constant<bool, false>;
  tile(HASNOT{}, 0, pfa, pfb);
  tile(HAS{}, 1, pfb, pfa);
  for (int t = 2; t < T; t += 2) {      // T = N / 64 is even (N % 128 == 0)
    tile(HAS{}, t, pfa, pfb);
    tile(HAS{}, t + 1, pfb, pfa);
  }
  // ---- tail: V(T−1) -> LDS, Oᵀ += Vᵀ(T−1)·Pᵀ(T−1)   (P of the last, odd tile = pfb)
#pragma unroll
  for (int i = 0; i < NPIECE; ++i) issue_v(i, T - 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  raw_barrier();
  rd0();
  static_for<NST>([&](auto stc) {
    pv_step(stc, pfb);
    __builtin_amdgcn_sched_barrier(0);
  });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  raw_barrier();   // every wave is done with V(T−1): the epilogue's staging aliases the tiles

  // ---- epilogue: O = Oᵀ / l through LDS (whole rows, 16-B stores).  Lane holds O[q = 16 qb + l16][d = 16 db + 4 g4 + (0..3)] in
  // a[4 (2 db + qb) ..]; every wave owns a private 32 x (ROWB + 16) B staging area (the KV tiles are dead).
  constexpr int ESTR = ROWB + 16;
  am_drain();
  float inv[2];
  inv[0] = 1.0f / an_x4_sum(l_run[0]);
  inv[1] = 1.0f / an_x4_sum(l_run[1]);
  char* stg = smem + wave * (32 * ESTR);
  // the lane id again, from mbcnt: keeping `lane` / `l16` / `g4` alive across the loop costs registers hipcc would spill
  const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int l16e = lane_e & 15, g4e = lane_e >> 4;
  static_for<NDB * 2>([&](auto ec) {
    constexpr int db = decltype(ec)::value >> 1, qb = decltype(ec)::value & 1;
    constexpr int base = 4 * (2 * db + qb);
    half4_t h;
    h[0] = cvt16<BF16>(am_acc_read<base + 0>() * inv[qb]);
    h[1] = cvt16<BF16>(am_acc_read<base + 1>() * inv[qb]);
    h[2] = cvt16<BF16>(am_acc_read<base + 2>() * inv[qb]);
    h[3] = cvt16<BF16>(am_acc_read<base + 3>() * inv[qb]);
    *(half4_t*)(stg + (16 * qb + l16e) * ESTR + (16 * db + 4 * g4e) * 2) = h;
  });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  half_t* ow = Ob + (size_t)q0 * D;
#pragma unroll
  for (int row = 0; row < 32; ++row) {       // one 1-KiB row per wave-instruction
    const u32x4_t v = *(const u32x4_t*)(stg + row * ESTR + lane_e * 16);
    *(u32x4_t*)(ow + (size_t)row * D + lane_e * 8) = v;
  }
}

}  // namespace lc
