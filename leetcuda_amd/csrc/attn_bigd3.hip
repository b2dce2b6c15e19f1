// attn_bigd3.hip — EXPERIMENTAL successor of attn_bigd2.hip for D = 256 / 512 (lc_tune_set "attn_d512" = 2; NOT the default,
// not yet run on a GPU: written at the end of round 2 after the GPU budget was spent, compiled and ISA-audited only).
//
// Why: attn_fwd_bigd2_kernel's waves wait 43 % of their cycles (profiles/r2N_pmc.json) — its K and V tiles of 64 rows are
// single-buffered, so every tile period has two barriers, each behind a vmcnt(0) on a tile DMA that was issued inside the
// same phase.  Here the KV tile is 32 rows and BOTH tiles are double-buffered (2 x (32 + 32) KiB at
// D = 512 = the 128 KiB attn_bigd2 uses for one K + one V tile):
//     tile period t (parity PAR = t & 1), ONE barrier:
//       Q·Kᵀ(t)      D/16 MFMAs from K slot PAR           | DMA: V(t) -> V slot PAR, K(t+1) -> K slot 1 − PAR (front-loaded)
//       P·V(t−1)     D/8 MFMAs from V slot 1 − PAR, softmax(t) as filler between its statements (attn_bigd2's phase B)
//       vmcnt(0), barrier   (K(t+1), V(t) published; everybody is done with K(t), V(t−1))
//   a DMA piece has a whole tile period (>= 2000 matrix-core cycles) of flight, the barrier cadence (one per 64 MFMAs of
//   32 cycles) is attn_w4u's.  Same wave tile (32 query rows x all D columns, Oᵀ in the 256 AGPRs), same fragment layouts
//   and swizzles as attn_bigd2 (tests/test_layouts.py), same "m is only a scale" softmax; Sᵀ is one 32 x 32 block (two
//   accumulation chains), P 8 registers per tile, so the Q fragments all stay in registers (no LDS parking).
// Reference: kernels/flash-attn/mma/basic/flash_attn_mma_tiling_qkv.cu:75-797, entry :881-945 (BASELINE config 5a).
#pragma once
#include "attn_bigd2.hip"

namespace lc {

constexpr int KVB3 = 32;   // KV rows per tile

// step 0's Vᵀ fragments of a V slot: eight transpose reads, in fragment order, into the fixed quads v[240:255]
template <int OFF, int HOFF>
LC_DEVINL void bd3_rd0(half8_t& f0, half8_t& f1, half8_t& f2, half8_t& f3, const uint32_t (&vx)[4]) {
  asm volatile("ds_read_b64_tr_b16 v[240:241], %4 offset:%8\n\tds_read_b64_tr_b16 v[242:243], %4 offset:%9\n\t"
               "ds_read_b64_tr_b16 v[244:245], %5 offset:%8\n\tds_read_b64_tr_b16 v[246:247], %5 offset:%9\n\t"
               "ds_read_b64_tr_b16 v[248:249], %6 offset:%8\n\tds_read_b64_tr_b16 v[250:251], %6 offset:%9\n\t"
               "ds_read_b64_tr_b16 v[252:253], %7 offset:%8\n\tds_read_b64_tr_b16 v[254:255], %7 offset:%9"
               : "={v[240:243]}"(f0), "={v[244:247]}"(f1), "={v[248:251]}"(f2), "={v[252:255]}"(f3)
               : "v"(vx[0]), "v"(vx[1]), "v"(vx[2]), "v"(vx[3]), "n"(OFF), "n"(OFF + HOFF));
}

template <int D>
constexpr int bigd3_lds_bytes() {   // 2 K slots + 2 V slots; the epilogue's O staging (4 waves x 32 rows x (2D + 16) B) aliases them
  constexpr int tiles = 4 * KVB3 * D * 2, stage = 4 * 32 * (2 * D + 16);
  return tiles > stage ? tiles : stage;
}

template <int D, bool BF16>
__global__ __launch_bounds__(256) void attn_fwd_bigd3_kernel(
    const half_t* __restrict__ Q, const half_t* __restrict__ K, const half_t* __restrict__ V,
    half_t* __restrict__ O, int N, int nqb, float sl2) {
  static_assert(D == 256 || D == 512, "bigd3: D = 256 or 512");
  constexpr int ROWB = D * 2;               // bytes per K / V row
  constexpr int TILE = KVB3 * ROWB;         // one K or V tile (32 KiB at D = 512)
  constexpr int NKS = D / 16;               // k-steps of Q·Kᵀ = MFMAs of the Q·Kᵀ phase
  constexpr int NDT = D / 32;               // 32-column Oᵀ blocks
  constexpr int CPR = ROWB / 16;            // 16-B chunks per row
  constexpr int PPR = ROWB / 1024;          // DMA pieces per row: 1 (D = 512); D = 256: one piece = 2 rows
  constexpr int NPIECE = TILE / 1024 / 4;   // DMA pieces per wave and tile (8 at D = 512, 4 at D = 256)
  static_assert(TILE + (NDT / 4 - 1) * 256 + 16 * ROWB + 8 * ROWB + 8 < 65536, "transpose-read offsets must fit 16 bits");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = wave_id();
  const int hi = lane >> 5, l32 = lane & 31;

  const int id = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x));
  const int bhi = __builtin_amdgcn_readfirstlane(id / nqb);
  const size_t bh = (size_t)bhi;
  const int q0 = __builtin_amdgcn_readfirstlane((id - bhi * nqb) * 128 + wave * 32);
  const half_t* Qb = Q + bh * (size_t)N * D;
  const char* Kb = (const char*)(K + bh * (size_t)N * D);
  const char* Vb = (const char*)(V + bh * (size_t)N * D);
  half_t* Ob = O + bh * (size_t)N * D;
  const int T = N / KVB3;                   // even (N % 128 == 0)
  const uint32_t smem32 = lds_addr32(smem);
  char* const ksm = smem;                   // K slots 0, 1
  char* const vsm = smem + 2 * TILE;        // V slots 0, 1

  // ---- LDS-DMA: as attn_bigd2 (piece = one 1-KiB row at D = 512, two 512-B rows at D = 256; this wave stages pieces
  // wave + 4 i; lane chunk slot cs holds source chunk cs ^ key(row))
  const buf_rsrc_t rk = make_rsrc(Kb), rv = make_rsrc(Vb);
  unsigned k_off[4], v_off;
  {
    const int cs = lane & (CPR - 1), rsub = PPR ? 0 : (lane >> 5);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = PPR ? (wave + 4 * j) : (2 * (wave + 4 * j) + rsub);
      k_off[j] = (unsigned)(rsub * ROWB + ((cs ^ (row & 15)) * 16));
    }
    const int rowv = PPR ? wave : (2 * wave + rsub);
    v_off = (unsigned)(rsub * ROWB + ((cs ^ ((rowv & 3) << 2)) * 16));
  }
  auto issue_k = [&](int i, int t) {   // piece i of tile t (clamped) -> K slot t & 1
    const int te = t < T ? t : T - 1;
    const int p = wave + 4 * i;
    blds16(rk, k_off[i & 3], (unsigned)te * TILE + (unsigned)p * 1024u, ksm + (t & 1) * TILE + p * 1024);
  };
  auto issue_v = [&](int i, int t) {   // piece i of tile t -> V slot t & 1
    const int te = t < T ? t : T - 1;
    const int p = wave + 4 * i;
    blds16(rv, v_off, (unsigned)te * TILE + (unsigned)p * 1024u, vsm + (t & 1) * TILE + p * 1024);
  };
#pragma unroll
  for (int i = 0; i < NPIECE; ++i) issue_k(i, 0);

  // ---- Q fragments -> registers (once): lane holds Q[q0 + l32][16 ks + 8 hi .. +8]
  half8_t qf[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) qf[ks] = *(const half8_t*)(Qb + (size_t)(q0 + l32) * D + 16 * ks + 8 * hi);
  static_for<D / 2>([&](auto r) { am_acc_zero<decltype(r)::value>(); });

  // ---- fragment read addresses (slot 0; the slot parity and the k-step / step offsets are immediates)
  const char* kx[8];   // K: row l32, chunk (2ks + hi): low 4 bits XOR (row & 15); + (ks >> 3) * 256 + slot * TILE as immediate
#pragma unroll
  for (int k8 = 0; k8 < 8; ++k8) kx[k8] = ksm + l32 * ROWB + (((2 * k8 + hi) ^ (l32 & 15)) * 16);
  const int vi = lane & 15, vgi = (lane >> 4) & 1;
  uint32_t vx[4];   // Vᵀ: kv row 4hi + (vi>>2) (+16g, +8), 64-B unit dt: low 2 bits XOR (row & 3)
#pragma unroll
  for (int b = 0; b < 4; ++b)
    vx[b] = smem32 + (uint32_t)(2 * TILE + (4 * hi + (vi >> 2)) * ROWB + 32 * vgi + 8 * (vi & 3) + ((b ^ (vi >> 2)) << 6));

  float m_run = -INFINITY, l_run = 0.f;
  half8_t pfa[2], pfb[2];   // P fragments (k-step g = 16 kv rows) of the even / odd tiles
  half8_t vf0, vf1, vf2, vf3;
  constexpr int NQ = NDT / 4, NST = 2 * NQ;   // P·V steps of four MFMAs: (g, dq), g = 0..1
  constexpr int VHOFF = 8 * ROWB;

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  raw_barrier();   // K(0) landed

  // P·V(t') step st from V slot VS with the P fragments pf
  auto pv_step = [&](auto stc, auto vsc, half8_t (&pf)[2]) {
    constexpr int st = decltype(stc)::value, VS = decltype(vsc)::value, g = st / NQ, dq = st % NQ;
    constexpr bool RD = st + 1 < NST;
    constexpr int g1 = (st + 1) / NQ, dq1 = (st + 1) % NQ;
    bd2_pv4_fix<64 * dq, BF16, RD, VS * TILE + dq1 * 256 + g1 * 16 * ROWB, VHOFF>(vf0, vf1, vf2, vf3, pf[g], vx);
  };
  auto rd0 = [&](auto vsc) {
    constexpr int VS = decltype(vsc)::value;
    bd3_rd0<VS * TILE, VHOFF>(vf0, vf1, vf2, vf3, vx);
  };

  // ---- one tile period.  pn = P(t) (written), po = P(t−1) (read).  HAS_PV = false: tile 0.
  auto tile = [&](auto parc, auto pvc, int t, half8_t (&pn)[2], half8_t (&po)[2]) {
    constexpr int PAR = decltype(parc)::value;
    constexpr bool HAS_PV = decltype(pvc)::value;
    f32x16_t s[2];   // two accumulation chains (even / odd k-steps) of the ONE 32 x 32 Sᵀ block; written by k-steps 0 / 1
    {
      half8_t kfr[3];
      auto ldk = [&](auto kc, auto rc) {
        constexpr int ks = decltype(kc)::value, r = decltype(rc)::value;
        kfr[r] = *(const half8_t*)(kx[ks & 7] + (ks >> 3) * 256 + PAR * TILE);
      };
      ldk(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
      ldk(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
      static_for<NKS>([&](auto kc) {
        constexpr int ks = decltype(kc)::value;
        if constexpr (ks + 2 < NKS) ldk(std::integral_constant<int, ks + 2>{}, std::integral_constant<int, (ks + 2) % 3>{});
        // both DMAs of this period in the first k-steps: V(t) -> V slot PAR, K(t+1) -> K slot 1 − PAR
        if constexpr (ks < NPIECE) {
          issue_v(ks, t);
          issue_k(ks, t + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        bd2_qk<BF16, (ks < 2)>(s[ks & 1], kfr[ks % 3], qf[ks]);
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    am_drain(s[0], s[1]);   // asm MFMAs: hipcc does not know their latency; VALU reads S next
#pragma unroll
    for (int r = 0; r < 16; ++r) s[0][r] += s[1][r];

    // ---- P·V(t−1) with softmax(t) as filler
    float ps0 = 0.f, ps1 = 0.f;
    const float nm = -m_run;
    if constexpr (HAS_PV) rd0(std::integral_constant<int, 1 - PAR>{});
    static_for<NST>([&](auto stc) {
      constexpr int st = decltype(stc)::value;
      if constexpr (HAS_PV) pv_step(stc, std::integral_constant<int, 1 - PAR>{}, po);
      __builtin_amdgcn_sched_barrier(0);
      static_for<16 / NST>([&](auto jc) {
        constexpr int r = st * (16 / NST) + decltype(jc)::value;
        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[0][r], sl2, nm));
        if constexpr ((r & 1) != 0) ps1 += p; else ps0 += p;
        pn[r >> 3][r & 7] = cvt16<BF16>(p);
      });
      __builtin_amdgcn_sched_barrier(0);
    });
    float psum = ps0 + ps1;
    if (!__all(psum_below(psum, 16384.0f)) || !HAS_PV) {        // overflow guard / first tile: establish the true max
      float mx = s[0][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[0][r]);
      mx = am_xhalf_max(mx * sl2);                   // (sl2 > 0)
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // exp2(-inf) = 0 on the first tile
      m_run = m_new;
      l_run *= alpha;
      am_drain();    // the P·V MFMAs of this period have written Oᵀ
      static_for<D / 2>([&](auto rc) { am_acc_scale<decltype(rc)::value>(alpha); });
      psum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[0][r], sl2, -m_run));
        psum += p;
        pn[r >> 3][r & 7] = cvt16<BF16>(p);
      }
    }
    l_run += psum;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // own V(t), K(t+1) pieces landed, own LDS reads retired
    raw_barrier();                                                // K(t), V(t−1) dead; K(t+1), V(t) complete
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  using HAS = std::integral_constant<bool, true>;
  using HASNOT = std::integral_constant<bool, false>;
  tile(P0{}, HASNOT{}, 0, pfa, pfb);
  tile(P1{}, HAS{}, 1, pfb, pfa);
  for (int t = 2; t < T; t += 2) {
    tile(P0{}, HAS{}, t, pfa, pfb);
    tile(P1{}, HAS{}, t + 1, pfb, pfa);
  }
  // ---- tail: Oᵀ += Vᵀ(T−1)·Pᵀ(T−1): V(T−1) sits in V slot 1 (published by the last barrier), P of the odd tile = pfb
  rd0(P1{});
  static_for<NST>([&](auto stc) {
    pv_step(stc, P1{}, pfb);
    __builtin_amdgcn_sched_barrier(0);
  });
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (the clamped K(T) pieces of the last period too)
  raw_barrier();   // every wave is done with the tiles: the epilogue's staging aliases them

  // ---- epilogue: O = Oᵀ / l through LDS (as attn_bigd2)
  constexpr int ESTR = ROWB + 16;
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
  half_t* ow = Ob + (size_t)q0 * D;
  constexpr int LPR = ROWB / 16;
  constexpr int RPI = 64 / LPR;
#pragma unroll
  for (int it = 0; it < 32 / RPI; ++it) {
    const int row = it * RPI + lane_e / LPR, c = lane_e % LPR;
    const u32x4_t v = *(const u32x4_t*)(stg + row * ESTR + c * 16);
    *(u32x4_t*)(ow + (size_t)row * D + c * 8) = v;
  }
}

}  // namespace lc
