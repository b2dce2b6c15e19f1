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
//   * KV tile = 32 rows: K tile 32 x 1024 and V tile 32 x 1024 single-buffered in LDS (64 KiB each), filled by LDS-DMA in the
//     shadow of the OTHER phase exactly as in attn_bigd2.hip;
//   * phase A: partial Sᵀ(t) = K(t)[:, my d-half] · Qᵀ[my d-half] — 32 MFMAs in two chains — is summed and written to a 4-KiB
//     exchange area, [barrier], read back from the partner and ADDED (a + b == b + a bit for bit: both members hold the same Sᵀ,
//     hence the same m, P and l — the softmax is computed redundantly, 16 exps per lane and tile, as filler);
//   * phase B: Oᵀ[my 512 columns] += Vᵀ(t−1)[my columns] · Pᵀ(t−1) — 32 MFMAs — with softmax(t) between the statements.
//   Per wave and tile: 64 MFMAs (= 2048 matrix-core cycles), 64 KiB of K / V fragment reads + 8 KiB of exchange, two barriers.
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

template <int SP8>   // the DMA pieces of a phase are spread over SP8 eighths of it (A/B knob, lc_tune_set "attn_d1024")
__global__ __launch_bounds__(256) void attn_fwd_bigd4_kernel(
    const half_t* __restrict__ Q, const half_t* __restrict__ K, const half_t* __restrict__ V,
    half_t* __restrict__ O, int N, int nqb, float sl2) {
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

  const int id = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x));
  const int bhi = __builtin_amdgcn_readfirstlane(id / nqb);
  const size_t bh = (size_t)bhi;
  const int q0 = __builtin_amdgcn_readfirstlane((id - bhi * nqb) * 64 + pair * 32);
  const int dcol = __builtin_amdgcn_readfirstlane(dhf * DH);                 // first column of this wave's d-half
  const half_t* Qb = Q + bh * (size_t)N * D;
  const char* Kb = (const char*)(K + bh * (size_t)N * D);
  const char* Vb = (const char*)(V + bh * (size_t)N * D);
  half_t* Ob = O + bh * (size_t)N * D;
  const int T = N / BD4_KVB;
  const uint32_t smem32 = lds_addr32(smem);
  char* const ksm = smem;
  char* const vsm = smem + TILE;

  // ---- LDS-DMA: piece i of this wave = half (i >> 3) of row wave + 4 (i & 7): 64 lanes x 16 B = 1 KiB.  Lane chunk slot cs holds
  // source chunk cs ^ key(row) (K: row & 15 — four values over i -> k_off[i & 3]; V: (row & 3) << 2 = (wave & 3) << 2), all inside
  // the lane's own 256-B group, so a half-row stays a half-row.
  const buf_rsrc_t rk = make_rsrc(Kb), rv = make_rsrc(Vb);
  unsigned k_off[4], v_off;
#pragma unroll
  for (int j = 0; j < 4; ++j) k_off[j] = (unsigned)(((lane & 63) ^ ((wave + 4 * j) & 15)) * 16);
  v_off = (unsigned)(((lane & 63) ^ ((wave & 3) << 2)) * 16);
  auto piece_off = [&](int i) { return (unsigned)((wave + 4 * (i & 7)) * ROWB + (i >> 3) * 1024); };
  auto issue_k = [&](int i, int t) {
    const int te = t < T ? t : T - 1;
    blds16(rk, k_off[i & 3], (unsigned)te * TILE + piece_off(i), ksm + piece_off(i));
  };
  auto issue_v = [&](int i, int t) {
    const int te = t < T ? t : T - 1;
    blds16(rv, v_off, (unsigned)te * TILE + piece_off(i), vsm + piece_off(i));
  };
#pragma unroll
  for (int i = 0; i < NPIECE; ++i) issue_k(i, 0);

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

  // ---- fragment read addresses (attn_bigd2.hip's layouts on 2-KiB rows; this wave's half starts 1 KiB into every row)
  const char* kx[8];   // K: row l32, chunk 64 half + 2 ks + hi: low 4 bits XOR (row & 15); + (ks >> 3) * 256 as immediate
#pragma unroll
  for (int k8 = 0; k8 < 8; ++k8) kx[k8] = ksm + l32 * ROWB + dhf * 1024 + (((2 * k8 + hi) ^ (l32 & 15)) * 16);
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

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  raw_barrier();   // K(0) landed; every wave's parked Q is its own (lane-private slots: no barrier needed for it)

  half8_t vf0, vf1, vf2, vf3;
  constexpr int NQ = NDT / 4, NST = 2 * NQ;       // P·V steps per tile: (g, dq), g = 0, 1 (16 kv rows each), dq = quad of d tiles
  constexpr int SPAN_A = SP8 * NKS / 8, SPAN_B = SP8 * NST / 8;   // DMA pieces spread over 7/8 of a phase: the texture-address unit (16 cycles per piece, four waves) is busy for the whole phase at full MFMA rate
  auto rd0 = [&]() { bd2_rd0<8 * ROWB>(vf0, vf1, vf2, vf3, vx); };
  auto pv_step = [&](auto stc, half8_t (&pf)[2]) {
    constexpr int st = decltype(stc)::value, g = st / NQ, dq = st % NQ;
    constexpr int g1 = (st + 1) / NQ, dq1 = (st + 1) % NQ;
    bd2_pv4_fix<64 * dq, BF16, (st + 1 < NST), dq1 * 256 + g1 * 16 * ROWB, 8 * ROWB>(vf0, vf1, vf2, vf3, pf[g], vx);
  };

  // ---- one tile period (attn_bigd2.hip's, + the exchange).  pn = P(t) (written), po = P(t−1) (read).  HAS_PV = false: tile 0.
  auto tile = [&](auto pvc, int t, half8_t (&pn)[2], half8_t (&po)[2]) {
    constexpr bool HAS_PV = decltype(pvc)::value;
    f32x16_t s[2];   // two independent accumulation chains over the even / odd k-steps of this wave's d-half
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[c][r] = 0.f;
    {
      half8_t kfr[3], qfr[3];
      auto ldk = [&](auto kc, auto rc) {
        constexpr int ks = decltype(kc)::value, r = decltype(rc)::value;
        kfr[r] = *(const half8_t*)(kx[ks & 7] + (ks >> 3) * 256);
        if constexpr (ks >= NRES) qfr[r] = *(const half8_t*)(qpark + (ks - NRES) * 1024);
      };
      ldk(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
      ldk(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
      static_for<NKS>([&](auto kc) {
        constexpr int ks = decltype(kc)::value;
        if constexpr (ks + 2 < NKS) ldk(std::integral_constant<int, ks + 2>{}, std::integral_constant<int, (ks + 2) % 3>{});
        if constexpr (HAS_PV)
          static_for<NPIECE>([&](auto ic) {
            if constexpr (decltype(ic)::value * SPAN_A / NPIECE == ks) issue_v(decltype(ic)::value, t - 1);
          });
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ks < NRES) bd2_qk<BF16, (ks < 2)>(s[ks & 1], kfr[ks % 3], qf[ks]);
        else bd2_qk<BF16>(s[ks & 1], kfr[ks % 3], qfr[ks % 3]);
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    am_drain(s[0], s[1]);   // asm MFMAs: hipcc does not know their latency; VALU reads S next
    f32x16_t sp;            // this wave's partial Sᵀ(t)
#pragma unroll
    for (int r = 0; r < 16; ++r) sp[r] = s[0][r] + s[1][r];
#pragma unroll
    for (int c = 0; c < 4; ++c) *(f32x4_t*)(xmine + c * 1024) = f32x4_t{sp[4 * c], sp[4 * c + 1], sp[4 * c + 2], sp[4 * c + 3]};
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // own V(t−1) pieces landed, own K reads retired, partial written
    raw_barrier();                                                // K(t) is dead, V(t−1) complete, both partials visible

    // =========================== phase B
    f32x16_t sf;            // the full Sᵀ(t) = mine + the partner's (commutative: bit-identical in both members)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const f32x4_t o4 = *(const f32x4_t*)(xpeer + c * 1024);
#pragma unroll
      for (int r = 0; r < 4; ++r) sf[4 * c + r] = sp[4 * c + r] + o4[r];
    }
    float ps0 = 0.f, ps1 = 0.f;
    const float nm = -m_run;
    if constexpr (HAS_PV) rd0();
    static_for<NST>([&](auto stc) {
      constexpr int st = decltype(stc)::value;
      static_for<NPIECE>([&](auto ic) {
        if constexpr (decltype(ic)::value * SPAN_B / NPIECE == st) issue_k(decltype(ic)::value, t + 1);
      });
      if constexpr (HAS_PV) pv_step(stc, po);
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
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // own K(t+1) pieces landed, own V reads retired
    raw_barrier();                                                // V(t−1) is dead, K(t+1) complete, the exchange area is free again
  };
  using HAS = std::integral_constant<bool, true>;
  using HASNOT = std::integral_constant<bool, false>;
  tile(HASNOT{}, 0, pfa, pfb);
  tile(HAS{}, 1, pfb, pfa);
  for (int t = 2; t < T; t += 2) {      // T = N / 32 is even (N % 64 == 0)
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
