// attn_bigd7.hip — FlashAttention-2 forward, D = 256 (fp16 / bf16): 64 query rows per wave on v_mfma_f32_16x16x32, KV tiles of 32 rows in
// two rings of four, ONE barrier per tile (round 4).
//
// Reference: kernels/flash-attn/mma/basic/flash_attn_mma_tiling_qkv.cu:75-797 (entry :881-945) and the D = 256 rows of every
// `*_tiling_qkv` / `*_share_kv` / `*_share_qkv` entry swept by flash_attn_mma.py at (1,48,8192,256).
//
// Why: attn_bigd2<256> is the D = 512 design at half width — a wave owns 32 query rows, so every K / V fragment it reads from LDS feeds ONE
// MFMA, a workgroup stages K / V for 128 query rows, and a phase is only 1024 matrix-core cycles long with a barrier at both ends: 0.42
// of peak where the D = 512 kernel reaches 0.50.  At D = 256 the Oᵀ accumulators of 64 query rows fill the 256 AGPRs exactly, so:
//   * a wave = 4 query blocks qb of 16 rows (a workgroup: 256 query rows); KV tile = 2 kv blocks kvb of 16 rows;
//   * every K fragment (kvb, ds) and every Vᵀ fragment (db) feeds FOUR MFMAs (one per query block): half the LDS bytes per FLOP of
//     attn_bigd2<256>, half the LDS-DMA bytes per FLOP, on the MFMA form that is 14 % cheaper per FLOP at the cap;
//   * a K or V tile is 16 KiB, so LDS holds a ring of FOUR K tiles and FOUR V tiles (128 KiB): K(t+3) and V(t+2) are requested in period
//     t and waited for with s_waitcnt vmcnt(8) at the end of period t + 1 — a full period of flight at least, where attn_bigd2 has half
//     a period —, and the only barrier is the one at the end of a period (it publishes K(t+2) / V(t+1) and frees the slots of K(t) /
//     V(t−1)): no barrier between Sᵀ = K·Qᵀ and P·V, the four waves drift freely inside a period; K(t+1) being published a whole
//     period ahead, its first fragments are read BEFORE the barrier and a period starts on operands already in registers;
//   * Sᵀ block (kvb, qb) = Σ_ds K fragment (kvb, ds) x Q fragment (qb, ds), ds = 0 .. 7: lane (l16, g4) holds S[q = 16 qb + l16][kv = 16 kvb +
//     4 g4 + r]; the Pᵀ operand is lane-local: P(qb) = pack(S(0, qb)[0..3], S(1, qb)[0..3]) (attn_bigd6's layout with one half-tile);
//   * ONE set of P registers: the probabilities replace the scores in place (fp32) as fillers behind the P·V MFMAs of the PREVIOUS
//     tile and are packed to fp16 behind the last of them, when the previous tile's P is dead.  That needs the decision "does the
//     running maximum still hold" BEFORE the first score is overwritten: a lane compares the maximum of its 8 scores per query row with
//     m_run + 8 (log2 units; p <= 256) — in the MFMA gaps of the second P·V statement —; if any lane of the wave fails (rare; tile 0 takes the
//     exact path by construction), the wave recomputes Sᵀ(t) behind the P·V statements (K(t) is still in its slot) and takes the exact
//     path — row maximum across the four lane groups, Oᵀ and l rescaled, then the probabilities.  The P·V statements stay straight-line
//     code: the Vᵀ quads carry asm-issued reads from one statement to the next, and hipcc copies such registers around a branch
//     (isa_audit.py rules R3 / R7 caught exactly that in the first version).  Mathematically the reference's online softmax with a
//     lazily updated maximum.  The fast path's three VALU instructions per score sit IN the gaps between the MFMAs of the P·V statements
//     (generated: tools/gen_attn_bigd7.py; written in C++ between two statements they ran while the matrix pipe idled: 0.53 busy);
//   * registers: Oᵀ 256 AGPRs; Q fragments of three query blocks 96, the fourth parked in LDS (see below); Sᵀ / p 32, P 16, K ring 16 + 8,
//     Vᵀ quads 16;
//   * Oᵀ block (db, qb) = a[16 db + 4 qb ..]: lane holds O[q = 16 qb + l16][d = 16 db + 4 g4 + r].
// LDS images: K row r (512 B = 32 chunks of 16 B): chunk c at (c & 16) | ((c ^ (r & 15)) & 15); V row r (16 pairs of 32 B): pair p at
// (p & 8) | ((p ^ key(r)) & 7), key(r) = ((r & 3) << 1) | ((r >> 2) & 1) — attn_bigd6's images on 512-byte rows (tests/test_layouts.py).
#pragma once
#include "attn_bigd6.hip"

namespace lc {

#include "attn_bigd7_stmts.inc"   // bd7_qk8f / bd7_pvf: the statements on the pinned score registers (tools/gen_attn_bigd7.py)

constexpr int BD7_ROWB = 512;                 // bytes per K / V row
constexpr int BD7_KVB = 32;                   // rows per KV tile
constexpr int BD7_TILE = BD7_KVB * BD7_ROWB;  // 16 KiB
constexpr int BD7_RING = 4;                   // K tiles and V tiles resident / in flight
constexpr int BD7_ESTR = BD7_ROWB + 16;       // epilogue staging row stride
constexpr int BD7_PARK = 4 * (256 / 32) * 1024;   // the Q fragments of query block 3 (8 d-steps x 1 KiB per wave): 32 KiB
constexpr int bd7_lds_bytes() { return 2 * BD7_RING * BD7_TILE + BD7_PARK; }   // 160 KiB; the epilogue's staging (132 KiB) aliases it

// VT: V handed over as [B,H,D,N] (the reference's *_swizzle_qkv entries, d <= 256): the rows of that tensor already are Vᵀ.  The V tile in
// LDS is [256 d][32 kv] (64-B rows; 16-B chunk c of row r at slot c ^ ((−(r >> 2)) & 3): conflict-free for the 4 x 16 lane groups of a
// ds_read_b128, tests/test_layouts.py), a Vᵀ fragment is ONE ds_read_b128 — lane (l16, g4) <- row 16 db + l16, kv 8 g4 .. + 7 — and the
// P operand must then hold kv 8 g4 + e in slot e: Sᵀ block kvb's row m stands for kv = 8 (m >> 2) + 4 kvb + (m & 3), i.e. the K fragment
// reads K tile row 8 (l16 >> 2) + 4 kvb + (l16 & 3) (K image key (row & 3) | ((row >> 3) & 3) << 2, injective on those row sets).
template <bool BF16, bool VT>
__global__ __launch_bounds__(256) void attn_fwd_bigd7_kernel(
    const half_t* __restrict__ Q, const half_t* __restrict__ K, const half_t* __restrict__ V,
    half_t* __restrict__ O, int N, int nqb, float sl2) {
  constexpr int D = 256;
  constexpr int ROWB = BD7_ROWB, TILE = BD7_TILE, RING = BD7_RING;
  constexpr int NDS = D / 32;              // d-steps of Q·Kᵀ (8)
  constexpr int NDB = D / 16;              // 16-column Oᵀ blocks (16)
  constexpr int NPIECE = TILE / 1024 / 4;  // DMA pieces per wave and tile (4): a piece = two rows
  constexpr float THR = 8.0f;              // a score may exceed the running maximum by this much (log2 units) before the slow path
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = wave_id();
  const int g4 = lane >> 4, l16 = lane & 15;

  const int id = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x));
  const int bhi = __builtin_amdgcn_readfirstlane(id / nqb);
  const size_t bh = (size_t)bhi;
  const int q0 = __builtin_amdgcn_readfirstlane((id - bhi * nqb) * 256 + wave * 64);
  const half_t* Qb = Q + bh * (size_t)N * D;
  const char* Kb = (const char*)(K + bh * (size_t)N * D);
  const char* Vb = (const char*)(V + bh * (size_t)N * D);   // (either layout: a head is N * D elements)
  half_t* Ob = O + bh * (size_t)N * D;
  const int T = N / BD7_KVB;               // a multiple of 4 (N % 128 == 0): the tile loop walks four ring slots per trip
  const uint32_t smem32 = lds_addr32(smem);
  char* const ksm = smem;                  // K ring: slot s at s * TILE
  char* const vsm = smem + RING * TILE;    // V ring

  // ---- LDS-DMA: piece = 1 KiB = rows 2 p, 2 p + 1 of a tile; this wave stages pieces p = wave + 4 i.  Lane -> row b = lane >> 5 of the
  // piece, chunk slot cs = lane & 31, which holds source chunk (K) (cs & 16) | ((cs ^ (row & 15)) & 15), row & 15 = (2 wave + 8 i + b) & 15
  // -> k_off[i & 1]; (V) pair slot ps = cs >> 1 holds source pair (ps & 8) | ((ps ^ key(row)) & 7), key(row) = ((row & 3) << 1) |
  // ((row >> 2) & 1) = (((2 wave + b) & 3) << 1) | (wave >> 1) for every i -> one v_off
  const buf_rsrc_t rk = make_rsrc(Kb), rv = make_rsrc(Vb);
  unsigned k_off[4], v_off;
  {
    const int b = lane >> 5, cs = lane & 31, ps = cs >> 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // row of the lane in piece wave + 4 j: 2 wave + 8 j + b; key = row & 15, or (VT) (row & 3) | ((row >> 3) & 3) << 2 = ((2 wave + b) & 3) | j << 2
      const int key = VT ? (((2 * wave + b) & 3) | (j << 2)) : ((2 * wave + 8 * j + b) & 15);
      k_off[j] = (unsigned)(b * ROWB + (((cs & 16) | ((cs ^ key) & 15)) * 16));
    }
    if constexpr (VT) {
      // Vᵀ tile: piece p = d rows 16 p .. + 15 (64 B each): lane -> row lane >> 2, chunk slot lane & 3 <- source chunk slot ^ ((−(row >> 2)) & 3),
      // (row >> 2) & 3 = (lane >> 4) & 3 for every piece; source row stride = 2 N bytes
      v_off = (unsigned)(lane >> 2) * (unsigned)N * 2u + (unsigned)((((lane & 3) ^ ((0 - ((lane >> 4) & 3)) & 3))) * 16);
    } else {
      const int key = (((2 * wave + b) & 3) << 1) | (wave >> 1);
      v_off = (unsigned)(b * ROWB + (((((ps & 8) | ((ps ^ key) & 7)) << 1) | (cs & 1)) * 16));
    }
  }
  auto issue_k = [&](int i, int t, int slot) {
    const int te = t < T ? t : T - 1;
    const int p = wave + 4 * i;
    blds16(rk, k_off[i], (unsigned)te * TILE + (unsigned)p * 1024u, ksm + slot * TILE + p * 1024);
  };
  auto issue_v = [&](int i, int t, int slot) {
    const int te = t < 0 ? 0 : (t < T ? t : T - 1);
    const int p = wave + 4 * i;
    if constexpr (VT) blds16(rv, v_off, (unsigned)p * 32u * (unsigned)N + (unsigned)te * 64u, vsm + slot * TILE + p * 1024);
    else blds16(rv, v_off, (unsigned)te * TILE + (unsigned)p * 1024u, vsm + slot * TILE + p * 1024);
  };
  // virtual periods −3 .. −1: K(0..2), V(−1 (a dummy: tile 0 into slot 3), 0, 1) — every period issues 4 + 4 pieces, so that
  // "all but the youngest 8 pieces have landed" always means "everything requested before the period that just ended"
#pragma unroll
  for (int u = 0; u < 3; ++u) {
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) issue_k(i, u, u);
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) issue_v(i, u - 1, (u + 3) & 3);
  }

  // ---- Q fragments (once): lane holds Q[q0 + 16 qb + l16][32 ds + 8 g4 .. +8] — query blocks 0 .. 2 in registers (96), query block 3
  // parked in the 32 KiB of LDS the rings leave free (lane-private 16-B slots, + 1 KiB per d-step) and read back through a two-slot
  // register ring next to the K fragments: all four resident would need 128 + 130 registers
  half8_t qf[NDS][3];
#pragma unroll
  for (int ds = 0; ds < NDS; ++ds)
#pragma unroll
    for (int qb = 0; qb < 3; ++qb) qf[ds][qb] = *(const half8_t*)(Qb + (size_t)min(q0 + 16 * qb + l16, N - 1) * D + 32 * ds + 8 * g4);   // (row clamp: N % 256 == 128, see the epilogue)
  char* const qpark = smem + 2 * RING * TILE + wave * (NDS * 1024) + lane * 16;
#pragma unroll
  for (int ds = 0; ds < NDS; ++ds) *(half8_t*)(qpark + ds * 1024) = *(const half8_t*)(Qb + (size_t)min(q0 + 48 + l16, N - 1) * D + 32 * ds + 8 * g4);
  static_for<256>([&](auto r) { am_acc_zero<decltype(r)::value>(); });

  // ---- fragment read addresses (slot, ds >> 2, kvb / db >> 3 and the second kv block go into the immediate offsets)
  // K: Sᵀ row l16 of block kvb <- K tile row 16 kvb + l16, or (VT) 8 (l16 >> 2) + 4 kvb + (l16 & 3); chunk 4 ds + g4: low 4 bits XOR key(row) = l16
  // in both images
  constexpr int KVB1 = (VT ? 4 : 16) * ROWB;   // from kv block 0's row to kv block 1's
  const char* kx[4];
#pragma unroll
  for (int k4 = 0; k4 < 4; ++k4) kx[k4] = ksm + (VT ? 8 * (l16 >> 2) + (l16 & 3) : l16) * ROWB + (((4 * k4 + g4) ^ l16) * 16);
  // Vᵀ: kv row 4 g4 + (l16 >> 2) (+ 16: second transpose read), 8 bytes at column 4 (l16 & 3) of pair db: slot ((db & 7) ^ key) + (db & 8)
  // (VT: ONE address — row l16 of the Vᵀ tile, chunk g4 at slot g4 ^ ((−(l16 >> 2)) & 3); + 1024 per column block db as immediate)
  uint32_t vx[8];
#pragma unroll
  for (int b = 0; b < 8; ++b)
    vx[b] = VT ? smem32 + (uint32_t)(RING * TILE + l16 * 64 + ((g4 ^ ((0 - (l16 >> 2)) & 3)) * 16))
               : smem32 + (uint32_t)(RING * TILE + (4 * g4 + (l16 >> 2)) * ROWB + 8 * (l16 & 3) + ((b ^ (((l16 >> 2) << 1) | (g4 & 1))) * 32));
  // the two lane addresses and the immediates (OFF, HOFF) of the reads a P·V statement issues for step s1 (fragments db = 4 s1 + 2 hq, + 1)
  constexpr int V_HOFF = VT ? 1024 : 16 * ROWB;

  float m_run[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, l_run[4] = {0.f, 0.f, 0.f, 0.f};
  half8_t pf[4];   // P fragments [qb] of the previous tile
#pragma unroll
  for (int qb = 0; qb < 4; ++qb) pf[qb] = half8_t{};

  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  raw_barrier();   // K(0), K(1), V(0) landed

  half8_t vf0, vf1, vf2, vf3;
  // the K fragments (and the parked Q fragment) of d-step 0 of the NEXT tile are read before the barrier that ends a period — K(t+1) was
  // published one barrier earlier (see the wait below) —, so a period starts on operands that are already in registers
  half8_t knx[2], qnx;
  auto prefetch = [&](int kslot_bytes) {
    knx[0] = *(const half8_t*)(kx[0] + kslot_bytes);
    knx[1] = *(const half8_t*)(kx[0] + kslot_bytes + KVB1);
    qnx = *(const half8_t*)(qpark);
  };
  prefetch(0);
  // ---- one period: Sᵀ(t) = K(t)·Qᵀ; then Oᵀ += Vᵀ(t−1)·Pᵀ(t−1) with softmax(t) as filler; K(t+3) / V(t+2) requested on the way; barrier.
  // SL = t & 3 (compile time: the ring slots are immediates).  HAS_PV = false: tile 0.
  auto period = [&](auto slc, auto pvc, int t) {
    constexpr int SL = decltype(slc)::value;
    constexpr bool HAS_PV = decltype(pvc)::value;
    constexpr int KS = SL * TILE, VS = ((SL + 3) & 3) * TILE;      // K(t), V(t−1)
    f32x4_t s[2][4];   // [kvb][qb]
    // Sᵀ(t) = K(t)·Qᵀ; DMA: with the K pieces of tile t + 3 (-> the slot K(t−1) left) behind every other d-step
    auto qk = [&](auto dmac) {
      constexpr bool DMA = decltype(dmac)::value;
      half8_t kfr[2][2], qr[2];   // K fragments / the parked Q fragment of d-step ds in ring slot ds & 1
      if constexpr (DMA) {        // (the first pass of a period: d-step 0 was prefetched)
        kfr[0][0] = knx[0];
        kfr[0][1] = knx[1];
        qr[0] = qnx;
      }
      auto ldk = [&](auto dc) {
        constexpr int ds = decltype(dc)::value, r = ds & 1;
        kfr[r][0] = *(const half8_t*)(kx[ds & 3] + KS + (ds >> 2) * 256);
        kfr[r][1] = *(const half8_t*)(kx[ds & 3] + KS + (ds >> 2) * 256 + KVB1);
        qr[r] = *(const half8_t*)(qpark + ds * 1024);
      };
      if constexpr (!DMA) ldk(std::integral_constant<int, 0>{});
      static_for<NDS>([&](auto dc) {
        constexpr int ds = decltype(dc)::value;
        if constexpr (ds + 1 < NDS) ldk(std::integral_constant<int, ds + 1>{});
        if constexpr (DMA && (ds & 1) == 0) issue_k(ds >> 1, t + 3, (SL + 3) & 3);
        __builtin_amdgcn_sched_barrier(0);
        bd7_qk8f<BF16, ds == 0>(s, kfr[ds & 1][0], kfr[ds & 1][1], qf[ds][0], qf[ds][1], qf[ds][2], qr[ds & 1]);
        // the Vᵀ fragments of the first P·V step: V(t−1) has been published for a whole period and the quads are idle during Sᵀ, so the
        // reads go out two d-steps before they are needed
        if constexpr (DMA && HAS_PV && ds == NDS - 3) bd7_rd<VT, VS, V_HOFF>(vf0, vf1, vf2, vf3, vx[0], vx[1], vx[2], vx[3]);
        __builtin_amdgcn_sched_barrier(0);
      });
    };
    // asm MFMAs: hipcc does not know their latency; VALU reads S next (the registers are operands of the drain: isa_audit.py rule R5)
    auto s_drain = [&]() {
      asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15"
                   : "+v"(s[0][0]), "+v"(s[0][1]), "+v"(s[0][2]), "+v"(s[0][3]), "+v"(s[1][0]), "+v"(s[1][1]), "+v"(s[1][2]), "+v"(s[1][3]));
    };
    // the exact path: row maximum across the four 16-lane groups (a row's kv columns are spread over them), Oᵀ and l rescaled to it, then
    // the probabilities from the raw scores
    float ps[4] = {0.f, 0.f, 0.f, 0.f};
    auto exact = [&]() {
      static_for<4>([&](auto qc) {
        constexpr int qb = decltype(qc)::value;
        const float mxl = fmaxf(fmaxf(fmaxf(s[0][qb][0], s[0][qb][1]), fmaxf(s[0][qb][2], s[0][qb][3])),
                                fmaxf(fmaxf(s[1][qb][0], s[1][qb][1]), fmaxf(s[1][qb][2], s[1][qb][3])));
        const float m_new = fmaxf(m_run[qb], an_x4_max(mxl * sl2));      // (sl2 > 0)
        const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_new);   // exp2(−inf) = 0 on the first tile
        m_run[qb] = m_new;
        l_run[qb] *= alpha;
        static_for<NDB>([&](auto dc) {
          static_for<4>([&](auto rc) { am_acc_scale<16 * decltype(dc)::value + 4 * qb + decltype(rc)::value>(alpha); });
        });
        ps[qb] = 0.f;
#pragma unroll
        for (int kvb = 0; kvb < 2; ++kvb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kvb][qb][r], sl2, -m_new));
            ps[qb] += p;
            s[kvb][qb][r] = p;
          }
      });
    };
    qk(std::integral_constant<bool, true>{});
    // ---- P·V(t−1): 4 steps x 2 statements in STRAIGHT-LINE code (no branch may separate two of them: the Vᵀ quads carry asm-issued reads
    // from one statement to the next); the V pieces of tile t + 2 (-> the slot V(t−2) left) behind every other statement
    auto pv_stmt = [&](auto xc) {
      constexpr int x = decltype(xc)::value, st = x >> 1, hq = x & 1, s1 = st + 1;
      constexpr int OFF = VT ? VS + (4 * s1 + 2 * hq) * 1024 : VS + (s1 >> 1) * 256;   // the reads it issues: step s1's fragments 2 hq, 2 hq + 1
      if constexpr ((x & 1) == 0) issue_v(x >> 1, t + 2, (SL + 2) & 3);
      if constexpr (HAS_PV) {
        const uint32_t ax = vx[VT ? 0 : 4 * (s1 & 1) + 2 * hq], ay = vx[VT ? 0 : 4 * (s1 & 1) + 2 * hq + 1];
        if constexpr (hq == 0) bd7_pvn<x, BF16, VT, 64 * st, OFF, V_HOFF>(vf0, vf1, pf[0], pf[1], pf[2], pf[3], ax, ay);
        else bd7_pvn<x, BF16, VT, 64 * st + 32, OFF, V_HOFF>(vf2, vf3, pf[0], pf[1], pf[2], pf[3], ax, ay);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    if constexpr (!HAS_PV) {
      // tile 0: nothing to accumulate yet and no running maximum: the exact path straight away
      static_for<8>([&](auto xc) { pv_stmt(xc); });   // (the V requests only)
      s_drain();
      exact();
    } else {
      pv_stmt(std::integral_constant<int, 0>{});
      // does the running maximum hold?  Per lane: the largest of its 8 raw scores of each query row against m_run + THR (log2 units) —
      // computed in the gaps of statement 1 (generated: bd7_pvc); the 8 MFMAs of statement 0 cover the latency of the Sᵀ MFMAs
      float over;
      {
        constexpr int OFF1 = VT ? VS + (4 + 2) * 1024 : VS;   // statement 1 issues step 1's fragments 2, 3
        bd7_pvc<BF16, VT, 32, OFF1, V_HOFF>(vf2, vf3, pf[0], pf[1], pf[2], pf[3], vx[VT ? 0 : 4 + 2], vx[VT ? 0 : 4 + 3], s, over, sl2,
                                            m_run[0], m_run[1], m_run[2], m_run[3]);
        __builtin_amdgcn_sched_barrier(0);
      }
      const bool hold = over <= THR;
      // p = exp2(s sl2 − m_run) in place, 32 per lane IN the MFMA gaps of the remaining six statements (generated: attn_bigd7_stmts.inc;
      // if the maximum does not hold the values are garbage, possibly inf, and are thrown away below)
      static_for<6>([&](auto xc) {
        constexpr int x = decltype(xc)::value + 2, st = x >> 1, hq = x & 1, s1 = st + 1;
        constexpr int OFF = VT ? VS + (4 * s1 + 2 * hq) * 1024 : VS + (s1 >> 1) * 256;
        constexpr int ta = bd7_pvf_ta(x), tb = bd7_pvf_tb(x);
        if constexpr ((x & 1) == 0) issue_v(x >> 1, t + 2, (SL + 2) & 3);
        const uint32_t ax = vx[VT ? 0 : 4 * (s1 & 1) + 2 * hq], ay = vx[VT ? 0 : 4 * (s1 & 1) + 2 * hq + 1];
        if constexpr (hq == 0)
          bd7_pvf<x, BF16, VT, 64 * st, OFF, V_HOFF>(vf0, vf1, pf[0], pf[1], pf[2], pf[3], ax, ay, s[ta >> 2][ta & 3], s[tb >> 2][tb & 3],
                                                     ps[ta & 3], ps[tb & 3], sl2, m_run[ta & 3], m_run[tb & 3]);
        else
          bd7_pvf<x, BF16, VT, 64 * st + 32, OFF, V_HOFF>(vf2, vf3, pf[0], pf[1], pf[2], pf[3], ax, ay, s[ta >> 2][ta & 3], s[tb >> 2][tb & 3],
                                                          ps[ta & 3], ps[tb & 3], sl2, m_run[ta & 3], m_run[tb & 3]);
        __builtin_amdgcn_sched_barrier(0);
      });
      if (!__all(hold)) {
        // rare (a row's maximum grew by more than 2^THR within one tile): the scores are gone — K(t) is still in its slot (it dies at the
        // barrier below), so Sᵀ(t) is computed again; P·V(t−1) has been issued completely and belongs to the OLD maximum: drain, rescale
        qk(std::integral_constant<bool, false>{});
        s_drain();
        am_drain();
        exact();
      }
    }
    // P(t−1) is dead (every P·V statement has been issued): pack P(t), row sums from the unrounded p (tiling_qkv.cu's order)
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) {
      l_run[qb] += ps[qb];
#pragma unroll
      for (int kvb = 0; kvb < 2; ++kvb)
#pragma unroll
        for (int r = 0; r < 4; ++r) pf[qb][4 * kvb + r] = cvt16<BF16>(s[kvb][qb][r]);
    }
    // d-step 0 of the next tile (its K tile was published by the previous barrier), then: everything requested before this period has
    // landed (K(t+2), V(t+1): all but the 8 youngest pieces), own K(t) / V(t−1) reads retired
    prefetch(((SL + 1) & 3) * TILE);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    raw_barrier();
  };
  using HAS = std::integral_constant<bool, true>;
  using HASNOT = std::integral_constant<bool, false>;
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  using S3 = std::integral_constant<int, 3>;
  period(S0{}, HASNOT{}, 0);
  period(S1{}, HAS{}, 1);
  period(S2{}, HAS{}, 2);
  period(S3{}, HAS{}, 3);
  for (int t = 4; t < T; t += 4) {
    period(S0{}, HAS{}, t);
    period(S1{}, HAS{}, t + 1);
    period(S2{}, HAS{}, t + 2);
    period(S3{}, HAS{}, t + 3);
  }
  // ---- tail: Oᵀ += Vᵀ(T−1)·Pᵀ(T−1): V(T−1) sits in slot (T − 1) & 3 = 3, published by the last barrier
  {
    constexpr int VS = 3 * TILE;
    bd7_rd<VT, VS, V_HOFF>(vf0, vf1, vf2, vf3, vx[0], vx[1], vx[2], vx[3]);
    static_for<8>([&](auto xc) {
      constexpr int x = decltype(xc)::value, st = x >> 1, hq = x & 1, s1 = st + 1;
      constexpr int OFF = VT ? VS + (4 * s1 + 2 * hq) * 1024 : VS + (s1 >> 1) * 256;
      const uint32_t ax = vx[VT ? 0 : 4 * (s1 & 1) + 2 * hq], ay = vx[VT ? 0 : 4 * (s1 & 1) + 2 * hq + 1];
      if constexpr (hq == 0) bd7_pvn<x, BF16, VT, 64 * st, OFF, V_HOFF>(vf0, vf1, pf[0], pf[1], pf[2], pf[3], ax, ay);
      else bd7_pvn<x, BF16, VT, 64 * st + 32, OFF, V_HOFF>(vf2, vf3, pf[0], pf[1], pf[2], pf[3], ax, ay);
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the clamped requests of the last periods have landed too
  raw_barrier();   // every wave is done with the rings: the epilogue's staging aliases them

  // ---- epilogue: O = Oᵀ / l through LDS (whole rows, 16-B stores).  Lane holds O[q = 16 qb + l16][d = 16 db + 4 g4 + (0..3)] in
  // a[16 db + 4 qb ..]; every wave owns a private 64 x (ROWB + 16) B staging area.
  constexpr int ESTR = BD7_ESTR;
  am_drain();
  float inv[4];
#pragma unroll
  for (int qb = 0; qb < 4; ++qb) inv[qb] = 1.0f / an_x4_sum(l_run[qb]);
  char* stg = smem + wave * (64 * ESTR);
  const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int l16e = lane_e & 15, g4e = lane_e >> 4;
  static_for<NDB * 4>([&](auto ec) {
    constexpr int db = decltype(ec)::value >> 2, qb = decltype(ec)::value & 3;
    constexpr int base = 16 * db + 4 * qb;
    half4_t h;
    h[0] = cvt16<BF16>(am_acc_read<base + 0>() * inv[qb]);
    h[1] = cvt16<BF16>(am_acc_read<base + 1>() * inv[qb]);
    h[2] = cvt16<BF16>(am_acc_read<base + 2>() * inv[qb]);
    h[3] = cvt16<BF16>(am_acc_read<base + 3>() * inv[qb]);
    *(half4_t*)(stg + (16 * qb + l16e) * ESTR + (16 * db + 4 * g4e) * 2) = h;
  });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  half_t* ow = Ob + (size_t)q0 * D;
  // N % 256 == 128 (round 5; legal in the reference, flash_attn_mma_share_qkv.cu:839): the head's last 256-row block has 128 real rows; its
  // waves 2 / 3 walked the KV tiles on a clamped copy of row N - 1 and store nothing (wave-uniform)
  if (q0 < N) {
#pragma unroll
    for (int it = 0; it < 32; ++it) {       // two 512-B rows per wave-instruction
      const int row = 2 * it + (lane_e >> 5);
      const u32x4_t v = *(const u32x4_t*)(stg + row * ESTR + (lane_e & 31) * 16);
      *(u32x4_t*)(ow + (size_t)row * D + (lane_e & 31) * 8) = v;
    }
  }
}

}  // namespace lc
