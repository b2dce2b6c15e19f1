// attn_bigd4.hip — FlashAttention-2 forward for D = 1024 (fp16): TWO waves share each block of 32 query rows, splitting the
// head dim between them (round 4; replaces the round-1 d-slice kernel attn_fwd_bigd_kernel for this head dim).
//
// Reference: kernels/flash-attn/mma/basic/flash_attn_mma_tiling_qkv.cu:75-797 (fine-grained d tiling, O(1) SRAM in D); its
// dispatcher and tiling_qk's reach d = 1024 (:904-910, :931-937; flash_attn_mma_tiling_qk.cu:862-...).
//
// Why a new structure: Oᵀ for 32 query rows x 1024 columns is 512 fp32 registers per lane — a whole SIMD's register file, twice
// what a wave can own — and round 1's answer (four workgroups per query block, each recomputing the full Q·Kᵀ for 256 output
// columns) does 2.5x the useful MFMA work at 135-194 TFLOP/s.  Here NOTHING is recomputed:
//   * a workgroup = 4 wave64 = 2 PAIRS, pair p = wave >> 1 owns query rows 32 p .. + 31 of the workgroup's 64; member h = wave & 1
//     owns the d-half [512 h, 512 h + 512): its Q fragments (32 rows x 512 d, 128 VGPR-equivalents as in attn_bigd2<512>), its half
//     of every Q·Kᵀ reduction and its 512 columns of Oᵀ (the 256 AGPRs);
//   * KV tile = 32 rows: K tile 32 x 1024 and V tile 32 x 1024 single-buffered in LDS (64 KiB each), filled by LDS-DMA with
//     HALF-TILE RECYCLING: each region is split by the order in which it is consumed (K: four d-panels, V: two row halves), a barrier
//     in the middle of each phase frees the half just read, and every batch of 8 pieces per wave is needed one and a half phases
//     after its issue (see "LDS images and LDS-DMA" below; the first version, with attn_bigd2's one-phase DMA, was TA + latency bound:
//     MFMA-busy 0.31, 720 - 745 TFLOP/s; this one 867);
//   * phase A: partial Sᵀ(t) = K(t)[:, my d-half] · Qᵀ[my d-half] — 32 MFMAs in two chains — is summed and written to a 4-KiB
//     exchange area, [barrier], read back from the partner and ADDED (a + b == b + a bit for bit: both members hold the same Sᵀ,
//     hence the same m, P and l — the softmax is computed redundantly, 16 exps per lane and tile, as filler);
//   * phase B: Oᵀ[my 512 columns] += Vᵀ(t−1)[my columns] · Pᵀ(t−1) — 32 MFMAs — with softmax(t) between the statements.
//   Per wave and tile: 64 MFMAs (= 2048 matrix-core cycles), 64 KiB of K / V fragment reads + 8 KiB of exchange, four barriers.
//   Per CU and tile the LDS-DMA moves 128 KiB for 64 query rows: 64 B/clk at full MFMA rate — twice attn_bigd2's and the whole rate
//   of the texture-address unit (one 1-KiB piece per 16 cycles).  That is what D = 1024 costs on this chip: 64 query rows per CU is
//   all the register file holds (64 x 1024 fp32 = half of a CU's registers), so a K / V byte feeds 128 FLOPs where D = 512 gets 256.
// LDS: K 64 KiB | V 64 KiB | exchange 16 KiB | parked Q fragments 16 KiB (4 k-steps per wave) = 160 KiB, a CU's whole LDS.
// Swizzles, the fixed V quads v[240:255], the stale-max softmax with its wave-uniform slow path, the Os2g epilogue: attn_bigd2.hip.
#pragma once
#include "attn_bigd2.hip"

namespace lc {

constexpr int BD4_KVB = 32;                 // KV rows per tile
constexpr int BD4_ROWB = 2048;              // bytes per K / V row (D = 1024)
constexpr int BD4_TILE = BD4_KVB * BD4_ROWB;   // 64 KiB
constexpr int BD4_PARK = 4;                 // Q k-steps parked in LDS per wave
constexpr int BD4_XCH = 2 * BD4_TILE;       // exchange area: 4 waves x 4 KiB
constexpr int BD4_QPK = BD4_XCH + 4 * 4096; // parked Q: 4 waves x PARK KiB
constexpr int BD4_LDS = BD4_QPK + 4 * BD4_PARK * 1024;
static_assert(BD4_LDS == 160 * 1024, "bigd4 uses a CU's whole LDS");
static_assert(2 * (512 / 32 / 4) == 8, "the softmax filler plan below is written for 8 P.V steps per tile");

// the eight transpose reads of a k-step's first P·V step (d tiles 0 .. 3; kv rows at OFF, second half at OFF + HOFF), in fragment
// order, into the fixed quads (bd2_rd0 with a row offset; a free function: clang refuses asm register constraints on lambda captures)
template <int OFF, int HOFF>
LC_DEVINL void bd4_rd_g(half8_t& f0, half8_t& f1, half8_t& f2, half8_t& f3, const uint32_t (&vx)[4]) {
  asm volatile("ds_read_b64_tr_b16 v[240:241], %4 offset:%8\n\tds_read_b64_tr_b16 v[242:243], %4 offset:%9\n\t"
               "ds_read_b64_tr_b16 v[244:245], %5 offset:%8\n\tds_read_b64_tr_b16 v[246:247], %5 offset:%9\n\t"
               "ds_read_b64_tr_b16 v[248:249], %6 offset:%8\n\tds_read_b64_tr_b16 v[250:251], %6 offset:%9\n\t"
               "ds_read_b64_tr_b16 v[252:253], %7 offset:%8\n\tds_read_b64_tr_b16 v[254:255], %7 offset:%9"
               : "={v[240:243]}"(f0), "={v[244:247]}"(f1), "={v[248:251]}"(f2), "={v[252:255]}"(f3)
               : "v"(vx[0]), "v"(vx[1]), "v"(vx[2]), "v"(vx[3]), "n"(OFF), "n"(OFF + HOFF));
}

template <int SP8>   // the DMA pieces of a phase are spread over SP8 eighths of it (A/B knob, lc_tune_set "attn_d1024")
__global__ __launch_bounds__(256) void attn_fwd_bigd4_kernel(
    const half_t* __restrict__ Q, const half_t* __restrict__ K, const half_t* __restrict__ V,
    half_t* __restrict__ O, int N, int nqb_arg, float sl2, int kv_stagger) {
  constexpr int D = 1024, DH = 512;        // head dim, the half a wave owns
  constexpr int ROWB = BD4_ROWB, TILE = BD4_TILE;
  constexpr int NKS = DH / 16;             // k-steps of this wave's half of Q·Kᵀ (32)
  constexpr int NDT = DH / 32;             // 32-column Oᵀ blocks of this wave (16)
  constexpr int NPIECE = TILE / 1024 / 4;  // DMA pieces per wave and tile (16): a piece = half a row
  constexpr bool BF16 = false;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = wave_id();
  const int hi = lane >> 5, l32 = lane & 31;
  const int pair = wave >> 1, dhf = wave & 1;   // pair of the workgroup, member of the pair (= which d-half)

  // nqb_arg < 0 (lc_tune_set "attn_bigd_map"): block b is query block b of the launch — consecutive blocks of a head go round-robin
  // over the 8 XCDs, so EVERY XCD streams the head's K / V for its share of the blocks: about twice the fabric bytes of the XCD-contiguous
  // map (nqb_arg > 0: one XCD owns consecutive blocks, bench.py attn_traffic_model), but all XCDs walk the same two heads, whose K / V
  // then live in the Infinity Cache.  Measured (profiles/r5f_bigd_map.log, r5f_bigd_map_pmc.log): D = 1024 + 3.7 % at 13.7 vs 7.2 GB
  // fetched, D = 512 - 2 % at 6.8 vs 2.0 GB — fabric bytes are not what bounds these kernels.  Same bits either way.
  const int nqb = nqb_arg < 0 ? -nqb_arg : nqb_arg;
  const int id = __builtin_amdgcn_readfirstlane(nqb_arg < 0 ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x));
  const int bhi = __builtin_amdgcn_readfirstlane(id / nqb);
  const size_t bh = (size_t)bhi;
  const int q0 = __builtin_amdgcn_readfirstlane((id - bhi * nqb) * 64 + pair * 32);
  const int dcol = __builtin_amdgcn_readfirstlane(dhf * DH);                 // first column of this wave's d-half
  const half_t* Qb = Q + bh * (size_t)N * D;
  const char* Kb = (const char*)(K + bh * (size_t)N * D);
  const char* Vb = (const char*)(V + bh * (size_t)N * D);
  half_t* Ob = O + bh * (size_t)N * D;
  const int T = N / BD4_KVB;
  // kv_stagger (lc_tune_set "attn_bigd_stagger"; auto: on with the round-robin block map): the workgroups of XCD x start their KV walk x
  // eighths of the sequence in and wrap — with the round-robin map all eight XCDs walk the same head, staggered they ask the fabric for
  // eight different tiles at a time instead of the same one (+ 1.8 ... 2.1 %, profiles/r5h_bigd_stagger.log; the GEMM's K stagger by XCD is the
  // same idea).  Softmax does not care about the order of the keys; the fp32 sums do: results agree to rounding, not bit for bit
  const int toff = kv_stagger ? __builtin_amdgcn_readfirstlane((int)(blockIdx.x & 7) * (T / 8)) : 0;
  const uint32_t smem32 = lds_addr32(smem);
  char* const ksm = smem;
  char* const vsm = smem + TILE;

  // ---- LDS images and LDS-DMA (round 4, second version: HALF-tile recycling).
  // K region = four PANELS of 16 KiB, panel P = 2 (d-half of the wave that reads it) + lohi: [32 rows][512 B] = the d range
  // [512 (P >> 1) + 256 (P & 1), + 256) of the tile's 32 rows; a DMA piece (1 KiB) = rows 2 p, 2 p + 1 of one panel (lanes 0-31 / 32-63),
  // 16-B chunk c of row r at slot c ^ (r & 15).  "K-lo" = panels 0, 2 (what the k-steps 0 .. 15 of the two d-halves read), "K-hi" =
  // panels 1, 3 (k-steps 16 .. 31).  V region = [32 rows][2048 B] row-major, 64-B unit u of row r at unit u ^ (r & 3); a piece = half
  // a row; "V-lo" = rows 0 .. 15 (P·V steps of k-step g = 0), "V-hi" = rows 16 .. 31 (g = 1).
  // Every half region is refilled as soon as ITS consumer half-phase is over (a barrier in the middle of each phase says so) and is
  // needed again one and a half phases later: a batch of 8 pieces per wave is issued in every half-phase —
  //     A(t) first half:  V-hi(t−1)      A(t) second half: K-lo(t+1)      B(t) first half:  K-hi(t+1)      B(t) second half: V-lo(t)
  // — and every synchronisation point waits for vmcnt(16): the batch it needs is the OLDEST of the three in flight (loads return
  // in order).  The first version of this kernel gave every piece one phase: the phase then lasted as long as the texture-address
  // unit needs for a whole tile (64 pieces x 16 cycles) plus an L2 read latency of ~700 cycles, MFMA-busy 0.31.
  const buf_rsrc_t rk = make_rsrc(Kb), rv = make_rsrc(Vb);
  unsigned k_off[4], v_off;
#pragma unroll
  for (int j = 0; j < 4; ++j) {   // this wave's K pieces: panel 2 (wave & 1) + lohi, row pairs p = (wave >> 1) + 2 i, i = 0 .. 7; key by i & 3
    const int row = 2 * ((wave >> 1) + 2 * j) + (lane >> 5);
    k_off[j] = (unsigned)((lane >> 5) * ROWB + (((lane & 31) ^ (row & 15)) * 16));
  }
  v_off = (unsigned)(((lane & 63) ^ ((wave & 3) << 2)) * 16);
  auto issue_k = [&](int i, int lohi, int t) {   // piece i (0 .. 7) of K-lo / K-hi of tile t (clamped)
    int te = (t < T ? t : T - 1) + toff;
    if (te >= T) te -= T;
    const int P = 2 * (wave & 1) + lohi, pp = (wave >> 1) + 2 * i;
    blds16(rk, k_off[i & 3], (unsigned)te * TILE + (unsigned)(2 * pp) * ROWB + (unsigned)P * 512u, ksm + P * 16384 + pp * 1024);
  };
  auto issue_v = [&](int i, int lohi, int t) {   // piece i (0 .. 7) of V-lo / V-hi of tile t: half (i >> 2) of row 16 lohi + wave + 4 (i & 3)
    int te = (t < T ? t : T - 1) + toff;
    if (te >= T) te -= T;
    const unsigned off = (unsigned)((16 * lohi + wave + 4 * (i & 3)) * ROWB + (i >> 2) * 1024);
    blds16(rv, v_off, (unsigned)te * TILE + off, vsm + off);
  };
#pragma unroll
  for (int i = 0; i < 8; ++i) issue_k(i, 0, 0);
#pragma unroll
  for (int i = 0; i < 8; ++i) issue_k(i, 1, 0);

  // ---- Q fragments of this wave's d-half -> registers (once): lane holds Q[q0 + l32][dcol + 16 ks + 8 hi .. +8]
  constexpr int PARK = BD4_PARK, NRES = NKS - PARK;
  half8_t qf[NRES];
#pragma unroll
  for (int ks = 0; ks < NRES; ++ks) qf[ks] = *(const half8_t*)(Qb + (size_t)(q0 + l32) * D + dcol + 16 * ks + 8 * hi);
  char* const qpark = smem + BD4_QPK + wave * (PARK * 1024) + lane * 16;   // + 1024 per parked k-step
#pragma unroll
  for (int i = 0; i < PARK; ++i)
    *(half8_t*)(qpark + i * 1024) = *(const half8_t*)(Qb + (size_t)(q0 + l32) * D + dcol + 16 * (NRES + i) + 8 * hi);
  static_for<DH / 2>([&](auto r) { am_acc_zero<decltype(r)::value>(); });

  // ---- fragment read addresses
  const char* kx[8];   // K: panel 2 dhf (+ ks >> 4 panels), row l32, chunk 16 ((ks >> 3) & 1) + 2 (ks & 7) + hi: low 4 bits XOR (row & 15)
#pragma unroll
  for (int k8 = 0; k8 < 8; ++k8) kx[k8] = ksm + dhf * 32768 + l32 * 512 + (((2 * k8 + hi) ^ (l32 & 15)) * 16);
  const int vi = lane & 15, vgi = (lane >> 4) & 1;
  uint32_t vx[4];      // Vᵀ: kv row 4 hi + (vi >> 2) (+ 16 g, + 8), 64-B unit dt: low 2 bits XOR (row & 3); + (dt >> 2) * 256 immediate
#pragma unroll
  for (int b = 0; b < 4; ++b)
    vx[b] = smem32 + (uint32_t)(TILE + (4 * hi + (vi >> 2)) * ROWB + dhf * 1024 + 32 * vgi + 8 * (vi & 3) + ((b ^ (vi >> 2)) << 6));
  // exchange: lane-linear 16-B slots, 4 per lane, 1 KiB apart
  char* const xmine = smem + BD4_XCH + wave * 4096 + lane * 16;
  const char* const xpeer = smem + BD4_XCH + (wave ^ 1) * 4096 + lane * 16;

  float m_run = -INFINITY, l_run = 0.f;
  half8_t pfa[2], pfb[2];   // P fragments (k-step g = 16 kv rows) of the even / odd tiles

  half8_t vf0, vf1, vf2, vf3;
  constexpr int NQ = NDT / 4, NST = 2 * NQ;       // P·V steps per tile: (g, dq), g = 0, 1 (16 kv rows each), dq = quad of d tiles
  constexpr int SPAN_AH = SP8 * (NKS / 2) / 8, SPAN_BH = SP8 * NQ / 8 > 0 ? SP8 * NQ / 8 : 1;   // a batch of 8 pieces over SP8 eighths of a HALF-phase
  // synchronisation point: the batch this half-phase reads has landed in every wave (it is the oldest of three in flight; the first
  // two tiles — fewer batches in flight — wait for everything), every wave is done with the half region the next batch refills
  auto sync_pt = [&](auto warmc) {
    if constexpr (decltype(warmc)::value) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
    raw_barrier();
  };
  // first Vᵀ fragments of k-step g (d tiles 0 .. 3) into the fixed quads
  auto rd_g = [&](auto gc) { bd4_rd_g<decltype(gc)::value * 16 * ROWB, 8 * ROWB>(vf0, vf1, vf2, vf3, vx); };
  // P·V step st = (g, dq): the four MFMAs of d tiles 4 dq .. + 3 and — inside a k-step — the transpose reads of the next step's
  // fragments; the last step of a k-step reads nothing (the other half of V is only known to have landed behind the barrier)
  auto pv_step = [&](auto stc, half8_t (&pf)[2]) {
    constexpr int st = decltype(stc)::value, g = st / NQ, dq = st % NQ;
    bd2_pv4_fix<64 * dq, BF16, (dq + 1 < NQ), (dq + 1) * 256 + g * 16 * ROWB, 8 * ROWB>(vf0, vf1, vf2, vf3, pf[g], vx);
  };

  // ---- one tile period.  pn = P(t) (written), po = P(t−1) (read).  HAS_PV = false: tile 0.  WARM: tiles 0 and 1.
  auto tile = [&](auto pvc, auto warmc, int t, half8_t (&pn)[2], half8_t (&po)[2]) {
    constexpr bool HAS_PV = decltype(pvc)::value;
    f32x16_t s[2];   // two independent accumulation chains over the even / odd k-steps of this wave's d-half; written by k-steps 0 / 1
    // =========================== phase A: Sᵀ(t) = K(t)[:, my d-half] · Qᵀ[my d-half]
    sync_pt(warmc);                                     // K-lo(t) landed; every wave is out of P·V(t−2 .. ): V-hi may be refilled
    static_for<2>([&](auto hc) {
      constexpr int KH = decltype(hc)::value, K0 = KH * (NKS / 2);
      if constexpr (KH == 1) sync_pt(warmc);            // K-hi(t) landed; K-lo(t) is dead
      half8_t kfr[3], qfr[3];
      auto ldk = [&](auto kc, auto rc) {
        constexpr int ks = decltype(kc)::value, r = decltype(rc)::value;
        kfr[r] = *(const half8_t*)(kx[ks & 7] + (ks >> 4) * 16384 + ((ks >> 3) & 1) * 256);
        if constexpr (ks >= NRES) qfr[r] = *(const half8_t*)(qpark + (ks - NRES) * 1024);
      };
      ldk(std::integral_constant<int, K0>{}, std::integral_constant<int, K0 % 3>{});
      ldk(std::integral_constant<int, K0 + 1>{}, std::integral_constant<int, (K0 + 1) % 3>{});
      static_for<NKS / 2>([&](auto kc) {
        constexpr int kl = decltype(kc)::value, ks = K0 + kl;
        if constexpr (kl + 2 < NKS / 2) ldk(std::integral_constant<int, ks + 2>{}, std::integral_constant<int, (ks + 2) % 3>{});
        static_for<8>([&](auto ic) {
          if constexpr (decltype(ic)::value * SPAN_AH / 8 == kl) {
            if constexpr (KH == 0) {
              if constexpr (HAS_PV) issue_v(decltype(ic)::value, 1, t - 1);   // V-hi(t−1)
            } else {
              issue_k(decltype(ic)::value, 0, t + 1);                         // K-lo(t+1)
            }
          }
        });
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ks < NRES) bd2_qk<BF16, (ks < 2)>(s[ks & 1], kfr[ks % 3], qf[ks]);
        else bd2_qk<BF16>(s[ks & 1], kfr[ks % 3], qfr[ks % 3]);
        __builtin_amdgcn_sched_barrier(0);
      });
    });
    am_drain(s[0], s[1]);   // asm MFMAs: hipcc does not know their latency; VALU reads S next
    f32x16_t sp;            // this wave's partial Sᵀ(t)
#pragma unroll
    for (int r = 0; r < 16; ++r) sp[r] = s[0][r] + s[1][r];
#pragma unroll
    for (int c = 0; c < 4; ++c) *(f32x4_t*)(xmine + c * 1024) = f32x4_t{sp[4 * c], sp[4 * c + 1], sp[4 * c + 2], sp[4 * c + 3]};
    sync_pt(warmc);                                     // V-lo(t−1) landed; K-hi(t) is dead; both partials visible

    // =========================== phase B: Oᵀ[my columns] += Vᵀ(t−1)[my columns] · Pᵀ(t−1), softmax(t) as filler
    f32x16_t sf;            // the full Sᵀ(t) = mine + the partner's (commutative: bit-identical in both members)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const f32x4_t o4 = *(const f32x4_t*)(xpeer + c * 1024);
#pragma unroll
      for (int r = 0; r < 4; ++r) sf[4 * c + r] = sp[4 * c + r] + o4[r];
    }
    float ps0 = 0.f, ps1 = 0.f;
    const float nm = -m_run;
    static_for<2>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      if constexpr (g == 1) sync_pt(warmc);             // V-hi(t−1) landed; V-lo(t−1) is dead
      if constexpr (HAS_PV) rd_g(gc);
      static_for<NQ>([&](auto dc) {
        constexpr int dq = decltype(dc)::value, st = g * NQ + dq;
        static_for<8>([&](auto ic) {
          if constexpr (decltype(ic)::value * SPAN_BH / 8 == dq) {
            if constexpr (g == 0) issue_k(decltype(ic)::value, 1, t + 1);     // K-hi(t+1)
            else issue_v(decltype(ic)::value, 0, t);                          // V-lo(t)
          }
        });
        if constexpr (HAS_PV) pv_step(std::integral_constant<int, st>{}, po);
        __builtin_amdgcn_sched_barrier(0);
        // softmax(t): row sums from the unrounded P (tiling_qkv.cu keeps the same order).  The 16 score elements of a lane ride behind
        // steps 2 .. 7 (3, 3, 3, 3, 2, 2): Sᵀ(t) is complete only when the partner's partial has come back from LDS, and a filler in
        // front of steps 0 / 1 would park the in-order stream on that read — the P·V MFMAs of those steps need only the Vᵀ fragments
        constexpr int E0 = st < 2 ? 0 : (st < 6 ? 3 * (st - 2) : 12 + 2 * (st - 6));
        constexpr int EN = st < 2 ? 0 : (st < 6 ? 3 : 2);
        static_for<EN>([&](auto jc) {
          constexpr int r = E0 + decltype(jc)::value;
          const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sf[r], sl2, nm));
          if constexpr ((r & 1) != 0) ps1 += p; else ps0 += p;
          pn[r >> 3][r & 7] = cvt16<BF16>(p);
        });
        __builtin_amdgcn_sched_barrier(0);
      });
    });
    float psum = ps0 + ps1;
    if (!__all(psum_below(psum, 16384.0f)) || !HAS_PV) {        // overflow guard / first tile: establish the true max
      float mx = sf[0];
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sf[r]);
      mx = am_xhalf_max(mx * sl2);                   // (sl2 > 0)
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // exp2(-inf) = 0 on the first tile
      m_run = m_new;
      l_run *= alpha;
      am_drain();    // the P·V MFMAs of this phase have written Oᵀ
      static_for<DH / 2>([&](auto rc) { am_acc_scale<decltype(rc)::value>(alpha); });
      psum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sf[r], sl2, -m_run));
        psum += p;
        pn[r >> 3][r & 7] = cvt16<BF16>(p);
      }
    }
    l_run += psum;
  };
  using HAS = std::integral_constant<bool, true>;
  using HASNOT = std::integral_constant<bool, false>;
  tile(HASNOT{}, HAS{}, 0, pfa, pfb);
  tile(HAS{}, HAS{}, 1, pfb, pfa);
  for (int t = 2; t < T; t += 2) {      // T = N / 32 is even (N % 64 == 0)
    tile(HAS{}, HASNOT{}, t, pfa, pfb);
    tile(HAS{}, HASNOT{}, t + 1, pfb, pfa);
  }
  // ---- tail: Oᵀ += Vᵀ(T−1)·Pᵀ(T−1)   (P of the last, odd tile = pfb).  V-lo(T−1) is in flight (or landed); V-hi(T−1) has not been asked for
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  raw_barrier();                        // every wave is out of P·V(T−2): V-hi may be refilled
#pragma unroll
  for (int i = 0; i < 8; ++i) issue_v(i, 1, T - 1);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // everything but the 8 pieces just issued: V-lo(T−1) landed
  raw_barrier();
  rd_g(std::integral_constant<int, 0>{});
  static_for<NQ>([&](auto dc) {
    pv_step(std::integral_constant<int, decltype(dc)::value>{}, pfb);
    __builtin_amdgcn_sched_barrier(0);
  });
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  raw_barrier();
  rd_g(std::integral_constant<int, 1>{});
  static_for<NQ>([&](auto dc) {
    pv_step(std::integral_constant<int, NQ + decltype(dc)::value>{}, pfb);
    __builtin_amdgcn_sched_barrier(0);
  });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  raw_barrier();   // every wave is done with V(T−1), every DMA piece has landed: the epilogue's staging aliases the tiles

  // ---- epilogue: this wave's 32 rows x 512 columns of O = Oᵀ / l through LDS (whole 1-KiB half-rows, 16-B stores).  Lane holds
  // O[q = l32][d = dcol + 32 dt + 8 rq + 4 hi + (0..3)] in a[16 dt + 4 rq ..]; every wave owns a private 32 x (1024 + 16) B staging area.
  constexpr int ESTR = DH * 2 + 16;
  am_drain();
  const float inv = 1.0f / am_xhalf_sum(l_run);
  char* stg = smem + wave * (32 * ESTR);
  const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int l32e = lane_e & 31, hie = lane_e >> 5;
  static_for<NDT * 4>([&](auto ec) {
    constexpr int dt = decltype(ec)::value >> 2, rq = decltype(ec)::value & 3;
    constexpr int base = 16 * dt + 4 * rq;
    half4_t h;
    h[0] = cvt16<BF16>(am_acc_read<base + 0>() * inv);
    h[1] = cvt16<BF16>(am_acc_read<base + 1>() * inv);
    h[2] = cvt16<BF16>(am_acc_read<base + 2>() * inv);
    h[3] = cvt16<BF16>(am_acc_read<base + 3>() * inv);
    *(half4_t*)(stg + l32e * ESTR + (32 * dt + 8 * rq + 4 * hie) * 2) = h;
  });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  half_t* ow = Ob + (size_t)q0 * D + dcol;
#pragma unroll
  for (int row = 0; row < 32; ++row) {       // one 1-KiB half-row per wave-instruction
    const u32x4_t v = *(const u32x4_t*)(stg + row * ESTR + lane_e * 16);
    *(u32x4_t*)(ow + (size_t)row * D + lane_e * 8) = v;
  }
}

}  // namespace lc
