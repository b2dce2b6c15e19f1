// attn_w8g.hip — FlashAttention-2 forward, D = 64: attn_w4g.hip's merged-phase kernel with EIGHT waves of 32 query rows — two waves per
// SIMD — instead of four waves of 64 (round 3; lc_tune_set "attn_nw" = 516).
//
// Same semantics / entry points as attn_fwd.hip (reference: kernels/flash-attn/mma/basic/flash_attn_mma_split_q.cu:55-699, dispatcher
// :769-815; the reference's published shapes (1,8,8192,64) / (1,48,8192,64), README.md:124-127).  Why: at D = 64 a phase carries 3.7
// softmax / LDS instructions per 16-cycle MFMA and ONE wave per SIMD issues an instruction per ~5.5 cycles: the MFMAs and the softmax of
// a wave barely overlap (MFMA-busy 0.47).  tools/attn_mix_probe.py (profiles/r3ab): MFMA + the softmax share of a slot cost 35.4 cycles
// per MFMA and SIMD with one wave per SIMD and 28.8 with two.  A 32-row wave needs Oᵀ 32 + K buffers 32 + Q~ 16 = 80 AGPRs and about
// 130 VGPRs: two waves fit a SIMD's 512 registers at D = 64 (not at D = 128: Oᵀ 64 + K 64 + Q 32).
// What changes against attn_w4g.hip: QB = 2 query blocks per wave (a phase = 16 MFMA slots, 8 softmax pairs, two slots per pair), the
// workgroup is 512 threads on the SAME 256-row query block (ring, tile size, grid and raster unchanged), one K and one V LDS-DMA
// piece per wave and tile, the AGPR map is packed into a[0:79] and every clobber list of this translation unit ends at a95
// (tu_attn_w8g.hip defines LC_AGPR_ALL).  Per query row nothing changes — the same MFMA k-order, the same exp2 / row-sum / pack
// sequence — so the outputs are BIT-IDENTICAL to attn_fwd_w4g_kernel<64> (GPU test).
#pragma once
#include "attn_w4g.hip"

namespace lc {

LC_DEVINL void w8_drain(f32x4_t (&s)[2][2]) {   // (am_drain for the [kvb][qb] blocks of a 32-row wave)
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(s[0][0]), "+v"(s[0][1]), "+v"(s[1][0]), "+v"(s[1][1])::"memory");
}

template <int D>
__global__ __launch_bounds__(512) void attn_fwd_w8g_kernel(
    const half_t* __restrict__ Q, const half_t* __restrict__ K, const half_t* __restrict__ V,
    half_t* __restrict__ O, int N, int nqb, float sl2) {
  static_assert(D == 64, "w8g attention kernel: D = 64 (the state of a 32-row wave fits 256 registers only there)");
  using G = W4G<D>;
  constexpr int NDS = G::NDS, NDB = G::NDB, ROWB = G::ROWB, TILE = G::TILE, SLOT = G::SLOT;
  constexpr int QB = 2;                        // 16-row query blocks per wave (attn_w4g: 4)
  constexpr int NS = 4 * QB * NDS;             // MFMA slots per phase: 2 QB NDS Q·Kᵀ alternating with QB NDB = 2 QB NDS P·V
  constexpr int NRV = G::NRV, NRK = G::NRK, KBUF = G::KBUF;
  constexpr int PPW = TILE / 1024 / 8;         // LDS-DMA pieces per wave and operand (eight waves)
  constexpr int GO = 0, GK = 4 * QB * NDB, GQ = GK + 2 * KBUF;   // AGPRs: Oᵀ blocks (db, qb) at 4 (QB db + qb), two K half-tile buffers, Q~ (qb, ds) at 4 (NDS qb + ds)
  static_assert(GQ + 4 * QB * NDS <= 96, "the AGPR clobber list of this translation unit ends at a95");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = wave_id();
  const int g4 = lane >> 4, l16 = lane & 15;

  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const size_t bh = id / nqb;
  const int q0 = (id - (int)bh * nqb) * 256 + wave * 32;
  const half_t* Qb = Q + bh * (size_t)N * D;
  const char* Kb = (const char*)(K + bh * (size_t)N * D);
  const char* Vb = (const char*)(V + bh * (size_t)N * D);
  half_t* Ob = O + bh * (size_t)N * D;
  const int T = N / KVB;
  const uint32_t smem32 = lds_addr32(smem);

  // ---- LDS-DMA: piece p = RPP rows x ROWB bytes; this wave stages pieces wave + 4 i (i = 0 .. PPW−1) of K and of V.
  // Lane -> row rr of the piece, 16-B slot cs of the row; the slot receives the logical chunk the read side expects there.
  unsigned k_off, v_off;
  if constexpr (D == 128) {
    const int rr = lane >> 4, cs = lane & 15;          // row & 15 = 4 (p & 3) + rr, p & 3 = wave
    k_off = (unsigned)(rr * 256 + ((cs ^ (4 * wave + rr)) * 16));
    v_off = (unsigned)(rr * 256 + (((((cs >> 1) ^ ((rr << 1) | (wave & 1))) << 1) | (cs & 1)) * 16));   // key: row & 3 = rr, (row >> 2) & 1 = wave & 1
  } else {
    const int rr = lane >> 3, cs = lane & 7;           // row & 15 = 8 (p & 1) + rr, p & 1 = wave & 1
    k_off = (unsigned)(rr * 128 + ((cs ^ (4 * (wave & 1) + (rr >> 1))) * 16));                          // (row >> 1) & 7
    v_off = (unsigned)(rr * 128 + (((((cs >> 1) ^ ((rr >> 1) & 3)) << 1) | (cs & 1)) * 16));            // key = (row >> 1) & 3 = (rr >> 1) & 3
  }
  const buf_rsrc_t rk = make_rsrc(Kb), rv = make_rsrc(Vb);
  auto issue_piece = [&](int i, int t) {   // i = 0 .. 2 PPW−1: K pieces, then V pieces; tile t (clamped) -> ring slot t & 3
    const int te = t < T ? t : T - 1;
    char* slot = smem + (t & 3) * SLOT;
    const int p = wave + 8 * (i % PPW);
    const unsigned so = (unsigned)te * TILE + (unsigned)p * 1024u;
    if (i < PPW)
      blds16(rk, k_off, so, slot + p * 1024);
    else
      blds16(rv, v_off, so, slot + TILE + p * 1024);
  };
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < 2 * PPW; ++i) issue_piece(i, t);

  // ---- Q~ = fp16(Q * scale*log2e) -> AGPRs: lane holds Q[q0 + 16 qb + l16][32 ds + 8 g4 .. +8]
  static_for<QB * NDS>([&](auto ic) {
    constexpr int i = decltype(ic)::value, qb = i / NDS, ds = i % NDS;
    const half8_t q = *(const half8_t*)(Qb + (size_t)(q0 + 16 * qb + l16) * D + 32 * ds + 8 * g4);
    half8_t qs;
#pragma unroll
    for (int e = 0; e < 8; ++e) qs[e] = (half_t)((float)q[e] * sl2);
    const u32x4_t w = __builtin_bit_cast(u32x4_t, qs);
    am_acc_write<GQ + 4 * i + 0>(w[0]);
    am_acc_write<GQ + 4 * i + 1>(w[1]);
    am_acc_write<GQ + 4 * i + 2>(w[2]);
    am_acc_write<GQ + 4 * i + 3>(w[3]);
  });
  static_for<4 * QB * NDB>([&](auto r) { am_acc_zero<decltype(r)::value>(); });

  // ---- fragment read offsets inside a ring slot
  uint32_t kx[NDS];   // K: row l16 (+16 kvb, +32 per half-tile: immediates), 16-B chunk (4 ds + g4) at its swizzled slot
#pragma unroll
  for (int ds = 0; ds < NDS; ++ds)
    kx[ds] = (uint32_t)(l16 * ROWB + (((4 * ds + g4) ^ (D == 128 ? l16 : ((l16 >> 1) & 7))) * 16));
  // Vᵀ transpose reads: kv row 4 g4 + (l16 >> 2) (+16 x, +32 per half-tile: immediates), 8 bytes at column 4 (l16 & 3) of pair db.
  // D = 128: vx[u] addresses pair 2 u (pair 2 u + 1 sits at ±32 B: key bit 0 = g4 & 1, not an immediate); D = 64: vx[db].
  constexpr int NVX = D == 128 ? 4 : NDB;
  uint32_t vx[NVX];
#pragma unroll
  for (int u = 0; u < NVX; ++u) {
    if constexpr (D == 128)
      vx[u] = (uint32_t)(TILE + (4 * g4 + (l16 >> 2)) * 256 + (((2 * u) ^ (((l16 >> 2) << 1) | (g4 & 1))) * 32) + 8 * (l16 & 3));
    else   // key(r) = (r >> 1) & 3 with r = 4 g4 + (l16 >> 2): ((g4 & 1) << 1) | (l16 >> 3)
      vx[u] = (uint32_t)(TILE + (4 * g4 + (l16 >> 2)) * 128 + ((u ^ (((g4 & 1) << 1) | (l16 >> 3))) * 32) + 8 * (l16 & 3));
  }
  const uint32_t vodd = (uint32_t)((g4 & 1) ? -32 : 32);

  uint32_t ka[NDS], vc[NVX], vp[NVX];   // this tile period's LDS addresses: K(t+1) fragments, Vᵀ of tile t / tile t−1
  auto set_tile_addrs = [&](int t) {
    const uint32_t sb_cur = smem32 + (uint32_t)((t & 3) * SLOT), sb_nxt = smem32 + (uint32_t)(((t + 1) & 3) * SLOT);
#pragma unroll
    for (int u = 0; u < NVX; ++u) {
      vp[u] = vc[u];
      vc[u] = vx[u] + sb_cur;
    }
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) ka[ds] = kx[ds] + sb_nxt;
  };
#pragma unroll
  for (int u = 0; u < NVX; ++u) vc[u] = vx[u] + smem32;

  f32x4_t sA[2][QB], sB[2][QB];   // Sᵀ blocks [kvb][qb] of the half-tile being exponentiated / being accumulated
  f32x4_t negm[QB];             // C operand of the first d-step: 4 x (−m) per query block
  half8_t pA[QB], pB[QB];       // P fragments [qb]
  half4_t vlo[NDB], vhi[NDB];   // Vᵀ fragments [db]: lo = kv block 0 rows, hi = kv block 1 rows (set A = db < NDB / 2, set B the rest)
  float l_run[QB] = {0.f, 0.f};

  auto read_k_all = [&](auto bufc, uint32_t sbase, auto hc) {   // (prologue only) one K half-tile -> AGPR buffer BUF
    constexpr int BUF = decltype(bufc)::value, H = decltype(hc)::value;
    static_for<NRK>([&](auto cc) {
      constexpr int c = decltype(cc)::value, kvb = c / NDS, ds = c % NDS;
      am_read_k<GK + KBUF * BUF + 4 * c, H * 32 * ROWB + kvb * 16 * ROWB>(kx[ds] + sbase);
    });
  };
  auto vaddr_of = [&](const uint32_t (&va)[NVX], int db) -> uint32_t {
    if constexpr (D == 128) return va[db >> 1] + ((db & 1) ? vodd : 0u);
    else return va[db];
  };
  auto wait_vset = [&](auto firstc) {   // retire the asm transpose reads of one Vᵀ set (NDB / 2 blocks from `first`)
    constexpr int first = decltype(firstc)::value;
    if constexpr (NDB == 8) am_wait_v8(reinterpret_cast<half4_t(&)[4]>(vlo[first]), reinterpret_cast<half4_t(&)[4]>(vhi[first]));
    else w4g_wait_v4(vlo[first], vlo[first + 1], vhi[first], vhi[first + 1]);
  };

  // ---- prologue: tiles 0, 1 landed; K(0), K(1) -> AGPR buffers 0, 1; Sᵀ(0), its row max, S − m, −m tuples
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  raw_barrier();
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using IB = std::integral_constant<int, NDB / 2>;
  read_k_all(I0{}, smem32, I0{});
  read_k_all(I1{}, smem32, I1{});
  am_lgkm0();
  static_for<2 * QB * NDS>([&](auto ic) {
    constexpr int i = decltype(ic)::value, ds = i >> 2, kvb = (i >> 1) & 1, qb = i & 1;
    if constexpr (ds == 0) an_qk_zero<GK + 4 * (NDS * kvb + ds), GQ + 4 * (NDS * qb + ds)>(sA[kvb][qb]);
    else an_qk<GK + 4 * (NDS * kvb + ds), GQ + 4 * (NDS * qb + ds)>(sA[kvb][qb]);
  });
  w8_drain(sA);
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
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
  auto phase = [&](auto hc, auto fc, int t, f32x4_t (&sr)[2][QB], f32x4_t (&sw)[2][QB], half8_t (&pw)[QB], half8_t (&pr)[QB]) {
    constexpr int H = decltype(hc)::value, F = decltype(fc)::value;
    constexpr bool HAS_PV = (F & 1) != 0, HAS_QK = (F & 2) != 0, HAS_KRD = (F & 4) != 0, HAS_DMA = (F & 8) != 0;
    constexpr int KQ = GK + KBUF * (1 - H);   // K(j+1) fragments: AGPR buffer (j+1) & 1
    constexpr int KRB = H;                     // K(j+2) goes to buffer (j+2) & 1 = H
    uint32_t (&vb_a)[NVX] = H == 0 ? vp : vc;  // Vᵀ(j−1) lives in tile t−1 (H = 0, its second half) or tile t (H = 1, first half)
    constexpr int VB_H = H == 0 ? 1 : 0;       // half-tile of Vᵀ(j−1) inside its tile
    // set A (read in slots NS/2 .. of the previous phase) and K(j+1) are needed from slot 0 / 1 on
    wait_vset(I0{});
    float ps[QB][2] = {{0.f, 0.f}, {0.f, 0.f}};
    float e0 = 0.f, e1 = 0.f, c0 = 0.f, c1 = 0.f;   // exps of the pair in flight / of the pair being packed
    auto pair_sum = [&](auto pc, auto wc, float a) {     // row sums from the unrounded P (split_q.cu:467-468)
      constexpr int qb = (decltype(pc)::value >> 1) & 1, w = decltype(wc)::value;
      ps[qb][w] += a;
      asm volatile("" : "+v"(ps[qb][w]));
    };
    auto pair_pack = [&](auto pc, float a, float b) {
      constexpr int p = decltype(pc)::value, kvb = p >> 2, qb = (p >> 1) & 1, k2 = p & 1;
      half2_t h2 = {(half_t)a, (half_t)b};
      asm volatile("" : "+v"(h2));
      pw[qb][4 * kvb + 2 * k2] = h2[0];
      pw[qb][4 * kvb + 2 * k2 + 1] = h2[1];
    };
    static_for<NS>([&](auto sc) {
      constexpr int s = decltype(sc)::value, i = s >> 1;
      // ---------------- the MFMA of this slot + the LDS reads in its shadow (one asm statement)
      // slots 0 .. NRV−1: Vᵀ(j−1) set B, transpose read c = s (db = NDB/2 + (c >> 1), kv block c & 1); slots 0 .. NRK−1: K(j+2)
      // fragment c = s; slots NS/2 .. NS/2 + NRV−1: Vᵀ(j) set A, transpose read c = s − NS/2 (db = c >> 1)
      constexpr bool RVB = s < NRV && HAS_PV, RK = s < NRK && HAS_KRD, RVA = s >= NS / 2 && s < NS / 2 + NRV;
      constexpr int RD = ((RVB || RVA) ? 1 : 0) | (RK ? 2 : 0);
      constexpr int c = RVA ? s - NS / 2 : (s % NRV), rdb = (RVA ? 0 : NDB / 2) + (c >> 1), rx = c & 1;
      constexpr int VOF = (RVA ? H : VB_H) * 32 * ROWB + rx * 16 * ROWB;
      half4_t& vout = rx ? vhi[rdb] : vlo[rdb];
      const uint32_t vaddr = RVA ? vaddr_of(vc, rdb) : vaddr_of(vb_a, rdb);
      constexpr int kc = s % NRK;
      constexpr int KR = GK + KBUF * KRB + 4 * kc, KOF = H * 32 * ROWB + (kc / NDS) * 16 * ROWB;
      if constexpr ((s & 1) == 0) {
        constexpr int ds = i >> 2, kvb = (i >> 1) & 1, qb = i & 1;
        constexpr int KIND = HAS_QK ? (ds == 0 ? 0 : 1) : 3;
        if constexpr (KIND != 3 || RD != 0)
          an_slot<KIND, RD, KQ + 4 * (NDS * kvb + ds), GQ + 4 * (NDS * qb + ds), VOF, KR, KOF>(
              sw[kvb][qb], negm[qb], half8_t{}, half8_t{}, vout, vaddr, ka[kc % NDS]);
      } else {
        constexpr int db = i >> 1, qb = i & 1;
        constexpr int KIND = HAS_PV ? 2 : 3;
        if constexpr (KIND != 3 || RD != 0)
          an_slot<KIND, RD, GO + 4 * (QB * db + qb), 0, VOF, KR, KOF>(sw[0][0], negm[0], cat4(vlo[db], vhi[db]), pr[qb], vout,
                                                                     vaddr, ka[kc % NDS]);
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- compiler-scheduled fillers behind it
      if constexpr (s == NS / 2 - 1 && HAS_PV) wait_vset(IB{});   // set B was read in slots 0 .. NRV−1; its first P·V MFMA is slot NS/2 + 1
      if constexpr (HAS_DMA && (s & 7) == 7) issue_piece(s >> 3, t + 2);   // 2 PPW pieces, one per 8 slots
      // softmax(j): 8 pairs (two consecutive kv of one query row) per phase, pair p -> kv block p >> 2, query block
      // (p >> 1) & 1, values 2 k2, 2 k2 + 1 of the 4; a pair is finished in pieces placed IN FRONT of a v_exp of the next pair
      // (hipcc pads a wait state between an asm statement and a transcendental that follows it directly)
      static_assert(NS == 16, "two slots per softmax pair");
      // two slots per pair: slot 2p: first row sum of pair p − 1, v_exp of pair p's first value; slot 2p + 1: second row sum and fp16 pack
      // of pair p − 1, v_exp of the second value
      constexpr int p = s >> 1, kvb = p >> 2, qb = (p >> 1) & 1, k2 = p & 1;
      if constexpr ((s & 1) == 0) {
        if constexpr (p >= 1) {
          pair_sum(std::integral_constant<int, p - 1>{}, I0{}, e0);
          c0 = e0;
          c1 = e1;
          __builtin_amdgcn_sched_barrier(0);
        }
        e0 = __builtin_amdgcn_exp2f(sr[kvb][qb][2 * k2]);
        asm volatile("" : "+v"(e0));
      } else {
        if constexpr (p >= 1) {
          pair_sum(std::integral_constant<int, p - 1>{}, I1{}, c1);
          pair_pack(std::integral_constant<int, p - 1>{}, c0, c1);
          __builtin_amdgcn_sched_barrier(0);
        }
        e1 = __builtin_amdgcn_exp2f(sr[kvb][qb][2 * k2 + 1]);
        asm volatile("" : "+v"(e1));
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    pair_sum(std::integral_constant<int, 4 * QB - 1>{}, I0{}, e0);
    pair_sum(std::integral_constant<int, 4 * QB - 1>{}, I1{}, e1);
    pair_pack(std::integral_constant<int, 4 * QB - 1>{}, e0, e1);
    // ---------------- overflow guard: m is only a scale; redo this half-tile with the true max when P got large
    // (bit patterns of non-negative floats order like unsigned integers, NaN / inf sit above every finite limit: ONE integer
    // compare of the largest pattern decides for the four query blocks — lc_common.h psum_below)
    uint32_t worst_bits = 0;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) worst_bits = max(worst_bits, __builtin_bit_cast(uint32_t, ps[qb][0] + ps[qb][1]));
    const bool ok = worst_bits < __builtin_bit_cast(uint32_t, AM_PSUM_LIMIT);
    if (!__all(ok)) {                                          // NaN / inf take this path too
      w8_drain(sw);                                            // every MFMA of this phase has written its result
      {
        float worst = 0.f;
        bool fin = true;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
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
      static_for<QB>([&](auto qc) {
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
        static_for<NDB>([&](auto dc) {
          constexpr int db = decltype(dc)::value;
          static_for<4>([&](auto rc) { am_acc_scale<GO + 4 * (QB * db + qb) + decltype(rc)::value>(alpha); });
        });
      });
      asm volatile("s_nop 3" ::: "memory");    // VALU writes of −m / S / P -> MFMA operand reads of the next phase
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) l_run[qb] += ps[qb][0] + ps[qb][1];
  };
  using F_FIRST0 = std::integral_constant<int, 2 | 4 | 8>;       // j = 0: no P·V(−1)
  using F_MID = std::integral_constant<int, 1 | 2 | 4 | 8>;
  using F_MID1 = std::integral_constant<int, 1 | 2 | 4>;         // odd phases do not issue DMA
  using F_LAST0 = std::integral_constant<int, 1 | 2>;            // j = 2T−2: no tile T to read K from / to stage
  using F_LAST1 = std::integral_constant<int, 1>;                // j = 2T−1: no Q·Kᵀ(2T)

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
    static_for<NRV>([&](auto cc) {
      constexpr int c = decltype(cc)::value, db = NDB / 2 + (c >> 1);
      if constexpr ((c & 1) == 0) vlo[db] = lds_tr16_asm<32 * ROWB>(vaddr_of(vc, db));
      else vhi[db] = lds_tr16_asm<32 * ROWB + 16 * ROWB>(vaddr_of(vc, db));
    });
    wait_vset(I0{});
    wait_vset(IB{});
    static_for<QB * NDB>([&](auto ic) {
      constexpr int i = decltype(ic)::value, db = i >> 1, qb = i & 1;
      an_pv<GO + 4 * (QB * db + qb)>(cat4(vlo[db], vhi[db]), pB[qb]);
    });
  }

  // ---- epilogue: O = Oᵀ / l through LDS (whole rows, 16-B stores).  Lane holds O[q = 16 qb + l16][d = 16 db + 4 g4 + (0..3)]
  // in a[4 (QB db + qb) ..]; every wave owns a private 32 x EPI_STRIDE staging area.
  am_drain();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  raw_barrier();                     // every wave is done with the KV ring
  float inv[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) inv[qb] = 1.0f / an_x4_sum(l_run[qb]);
  char* stg = smem + wave * (16 * QB * G::EPI_STRIDE);
  static_for<QB>([&](auto qc) {
    constexpr int qb = decltype(qc)::value;
    static_for<NDB>([&](auto dc) {
      constexpr int db = decltype(dc)::value;
      constexpr int base = GO + 4 * (QB * db + qb);
      half4_t h;
      h[0] = (half_t)(am_acc_read<base + 0>() * inv[qb]);
      h[1] = (half_t)(am_acc_read<base + 1>() * inv[qb]);
      h[2] = (half_t)(am_acc_read<base + 2>() * inv[qb]);
      h[3] = (half_t)(am_acc_read<base + 3>() * inv[qb]);
      *(half4_t*)(stg + (16 * qb + l16) * G::EPI_STRIDE + (16 * db + 4 * g4) * 2) = h;
    });
  });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private staging: own writes visible to own reads
  half_t* ow = Ob + (size_t)q0 * D;
  constexpr int LPR = ROWB / 16, RPI = 64 / LPR;      // lanes per row, rows per iteration
#pragma unroll
  for (int it = 0; it < 16 * QB / RPI; ++it) {
    const int row = it * RPI + lane / LPR;
    const u32x4_t v = *(const u32x4_t*)(stg + row * G::EPI_STRIDE + (lane % LPR) * 16);
    *(u32x4_t*)(ow + (size_t)row * D + (lane % LPR) * 8) = v;
  }
}

}  // namespace lc
