// attn_w4.hip — FlashAttention-2 forward, D = 128: FOUR wave64 per workgroup, 64 query rows per wave, ONE wave
// per SIMD with the whole 512-entry register file, software-pipelined inside one instruction stream.
//
// Same semantics / entry points as attn_fwd.hip (reference: kernels/flash-attn/mma/basic/
// flash_attn_mma_split_q.cu:55-699, flash_attn_mma_share_qkv.cu:46-769).
// Why: every 8-wave x 32-row schedule in attn_fwd.hip (lock-step, ping-pong, software-pipelined, four-cluster)
// lands on the same ~0.98 PFLOP/s — measured, the two waves of a SIMD do not hide each other's work, a segment
// costs the SUM of what both issue (an MFMA cluster next to a partner's softmax runs 1070 instead of 512
// cycles), every K / Vᵀ fragment (1 KiB of LDS per MFMA) serves only 32 query rows, and a KV tile needs 4-5
// workgroup barriers.  Here a fragment serves 64 rows (half the LDS reads per MFMA), there is one barrier per KV
// tile, and the only overlap relied on is the one that does work on this chip: independent VALU / LDS / DMA
// instructions in the issue shadow of the SAME wave's MFMAs (an MFMA issues in ~4 of its 32 cycles).
//
// Register plan (per lane; literal AGPR numbers, see lc_common.h LC_AGPR_ATTN):
//   a[0:127]    Oᵀ accumulators  O(qb, dt) = a[16(4qb+dt) ..]   qb = query block (rows q0+32qb+l32), dt = 32-col d tile
//   a[192:255]  Q fragments (MFMA B operand straight from AGPRs), Q(qb, ks) = a[192 + 4(8qb+ks) ..]
//   VGPRs       two Sᵀ buffers (2 x 64: the tile being exponentiated and the tile being accumulated, swapped every
//               iteration — the MFMA writes VGPRs directly, no accumulator copies), P fragments (32), and ONE block
//               of ~64 registers shared by the K and Vᵀ fragments (K fragments are read in two halves, the second
//               while the first is being consumed; a dead fragment's registers are reused by the next loads).
// Pipeline, iteration t (KV tile t), one barrier at its top:
//   phase 1: 32 MFMAs  Sᵀ(t+1) = K(t+1)·Qᵀ   | VALU: P(t) = exp2(E(t)), row sums, fp16 pack | LDS: Vᵀ(t) fragments
//            | DMA: tile t+3 (8 pieces, one per 4 MFMAs)
//   phase 2: 32 MFMAs  Oᵀ += Vᵀ(t)·Pᵀ(t)     | VALU: row max of Sᵀ(t+1), rescale decision, E(t+1) = S·scale − m
//            | LDS: K(t+2) fragments
//   Issue budget (tools/coissue_probe.py, one wave per SIMD): beside one 32-cycle MFMA the same wave issues for
//   free <= 4 plain VALU (5.6 cycles each alone) or <= 2 v_exp_f32 (8.8 each); a dependent use right behind a
//   v_exp stalls the stream (13 cycles per exp+add pair), packed-f32 VALU costs +8..18 cycles of MFMA time
//   each (not used), an LDS read +6..10.  Hence: plain fp32 math, the add / pack of a P pair one MFMA after its
//   exps, LDS reads one per MFMA.
//   (rare) O rescale after phase 2, when some row max grew by more than 2^RESCALE_THR (deferred rescale, as attn_fwd.hip).
// LDS: ring of 4 KV tiles (K 16 KiB + V 16 KiB each, unpadded 256-B rows, swizzles of attn_fwd_c4_kernel), staged by
// LDS-DMA.  Tile t+3 replaces tile t-1, whose last reads (Vᵀ(t-1), phase 1 of t-1) are complete before barrier(t);
// every wave waits for its own pieces of tile t+2 (issued in phase 1 of t-1) before barrier(t), after which Vᵀ(t),
// K(t+2) are readable by everybody.
#pragma once
#include "attn_fwd.hip"

namespace lc {

#define AW4_PRE "s_nop 1\n\t"   // VALU write -> MFMA operand read (hipcc reloads spilled operand pieces right in front)

constexpr int AW4_TILE = KVB * 128 * 2;     // 16 KiB: one K or V tile
constexpr int AW4_SLOT = 2 * AW4_TILE;      // K + V
constexpr int AW4_NSLOT = 4;
constexpr int AW4_LDS = AW4_NSLOT * AW4_SLOT;   // 128 KiB

// EMPIRICAL (bisected on hardware with per-statement drains, round 1): with Sᵀ accumulated in VGPRs (SrcC = vDst = arch VGPR tuple) the
// instruction stream right behind an ACCUMULATING Q·Kᵀ MFMA corrupts results unless >= 4 wait states follow it
// (s_nop 3 fixes, s_nop 7 used; the C = 0 form and the AGPR-accumulating P·V MFMAs need nothing; operands
// overwritten right after issue are safe — tools/mfma_war_probe.py).  Consistent with the documented XDL
// SrcC-read window (the MFMA is still reading its 16 SrcC VGPRs while the following VALU/TRANS op issues).
// (without it: ~1.04 PFLOP/s but wrong rows in the second query block, varying from launch to launch)
#define AW4_POSTQN "\n\ts_nop 7"
// Sᵀ block (VGPRs) (+)= K fragment x Q fragment (literal AGPRs);  ZERO: first k-step, C = 0.
// Hazards owned here (hipcc pads nothing around asm): the next MFMA on the same block comes 4 MFMAs later; VALU
// reads of a block start >= 3 MFMAs after its last write (phase 2 processes the blocks in write order); an
// accumulating MFMA is followed by s_nop 7 (AW4_POSTQN, empirical); every
// MFMA statement opens with s_nop 1 because under register pressure hipcc reloads pieces of the "v" operands from
// its AGPR spill slots (v_accvgpr_read) DIRECTLY in front of the statement — a VALU write -> MFMA operand read with
// zero wait states (observed: wrong P / K fragments in some rows, varying from launch to launch).
template <int QREG, bool ZERO>
LC_DEVINL void aw4_qk(f32x16_t& s, half8_t k) {
  if constexpr (ZERO)
    asm volatile(AW4_PRE "v_mfma_f32_32x32x16_f16 %0, %1, a[%2:%3], 0"
                 : "=&v"(s) : "v"(k), "n"(QREG), "n"(QREG + 3) : LC_AGPR_ATTN);
  else
    asm volatile(AW4_PRE "v_mfma_f32_32x32x16_f16 %0, %1, a[%2:%3], %0" AW4_POSTQN
                 : "+v"(s) : "v"(k), "n"(QREG), "n"(QREG + 3) : LC_AGPR_ATTN);
}
// Oᵀ(qb,dt) += Vᵀ fragment x Pᵀ fragment
template <int OACC>
LC_DEVINL void aw4_pv(half8_t v, half8_t p) {
  asm volatile(AW4_PRE "v_mfma_f32_32x32x16_f16 a[%2:%3], %0, %1, a[%2:%3]"
               :: "v"(v), "v"(p), "n"(OACC), "n"(OACC + 15) : LC_AGPR_ATTN);
}
template <int R>
LC_DEVINL float aw4_acc_read() {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(x) : "n"(R));
  return x;
}
template <int R>
LC_DEVINL void aw4_acc_write(uint32_t x) {
  asm volatile("v_accvgpr_write_b32 a[%1], %0" :: "v"(x), "n"(R) : LC_AGPR_ATTN);
}
template <int R>
LC_DEVINL void aw4_acc_zero() { asm volatile("v_accvgpr_write_b32 a[%0], 0" :: "n"(R) : LC_AGPR_ATTN); }
template <int R>
LC_DEVINL void aw4_acc_scale(float alpha) {   // a[R] *= alpha (rare path; MFMAs drained by the caller)
  float tmp;
  asm volatile("v_accvgpr_read_b32 %0, a[%2]\n\tv_mul_f32 %0, %0, %1\n\ts_nop 1\n\tv_accvgpr_write_b32 a[%2], %0"
               : "=&v"(tmp) : "v"(alpha), "n"(R) : LC_AGPR_ATTN);
}
LC_DEVINL void aw4_drain() { asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory"); }
// max over the two 32-lane halves (the lane's kv columns are split with lane ^ 32)
LC_DEVINL float aw4_xhalf_max(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return fmaxf(a, b);
}
LC_DEVINL float aw4_xhalf_sum(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}

// STAMPS (diagnosis, clobbers the first bytes of Q after they are in registers): s_memtime of wave 0 of workgroup 0 at
// [0] top of iteration (before the wait + barrier), [1] after the barrier, [2] after phase 1, [3] after phase 2,
// KV tiles 16..19.
template <int D, bool STAMPS = false>
__global__ __launch_bounds__(256) void attn_fwd_w4_kernel(
    const half_t* __restrict__ Q, const half_t* __restrict__ K, const half_t* __restrict__ V,
    half_t* __restrict__ O, int N, int nqb, float sl2) {
  static_assert(D == 128, "w4 attention kernel: D = 128 only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = wave_id();
  const int hi = lane >> 5, l32 = lane & 31;

  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const size_t bh = id / nqb;
  const int q0 = (id - (int)bh * nqb) * 256 + wave * 64;
  const half_t* Qb = Q + bh * (size_t)N * D;
  const char* Kb = (const char*)(K + bh * (size_t)N * D);
  const char* Vb = (const char*)(V + bh * (size_t)N * D);
  half_t* Ob = O + bh * (size_t)N * D;
  const int T = N / KVB;

  // ---- LDS-DMA: piece p = 4 rows x 256 B; this wave stages pieces wave + 4i (i = 0..3) of K and of V
  const int r4 = lane >> 4, cs = lane & 15;
  const unsigned k_off = (unsigned)(r4 * 256 + ((cs ^ (4 * wave + r4)) * 16));   // (row & 15) = 4(p&3) + r4, p&3 = wave
  const unsigned v_off = (unsigned)(r4 * 256 + ((cs ^ (r4 << 2)) * 16));         // (row & 3) = r4
  auto issue_piece = [&](int i, int t, int tslot = -1) {   // i = 0..7: K pieces, then V pieces; tile t -> ring slot
    const size_t tb = (size_t)t * AW4_TILE;
    char* slot = smem + ((tslot < 0 ? t : tslot) & 3) * AW4_SLOT;
    const int p = wave + 4 * (i & 3);
    if (i < 4)
      glds16(Kb + tb + (size_t)p * 1024 + k_off, slot + p * 1024);
    else
      glds16(Vb + tb + (size_t)p * 1024 + v_off, slot + AW4_TILE + p * 1024);
  };

  // prologue DMA: tiles 0, 1, 2
#pragma unroll
  for (int t = 0; t < 3; ++t)
    if (t < T) {
#pragma unroll
      for (int i = 0; i < 8; ++i) issue_piece(i, t);
    }

  // ---- Q fragments -> AGPRs: lane holds Q[q0 + 32qb + l32][16 ks + 8 hi .. +8]
  static_for<16>([&](auto ic) {
    constexpr int i = decltype(ic)::value, qb = i >> 3, ks = i & 7;
    const u32x4_t q = *(const u32x4_t*)(Qb + (size_t)(q0 + 32 * qb + l32) * D + 16 * ks + 8 * hi);
    aw4_acc_write<192 + 4 * i + 0>(q[0]);
    aw4_acc_write<192 + 4 * i + 1>(q[1]);
    aw4_acc_write<192 + 4 * i + 2>(q[2]);
    aw4_acc_write<192 + 4 * i + 3>(q[3]);
  });
  static_for<128>([&](auto r) { aw4_acc_zero<decltype(r)::value>(); });

  // ---- fragment read offsets (see attn_fwd_c4_kernel)
  const int k_rd = l32 * 256 + ((hi ^ (l32 & 15)) * 16);          // ^ (ks*32), + tt*8192
  const int vi = lane & 15, vgi = (lane >> 4) & 1;
  const int v_rd = AW4_TILE + (4 * hi + (vi >> 2)) * 256 + 32 * vgi + 8 * (vi & 3);
  const int v_sw = vi >> 2;

  half8_t kf[16];             // K fragments of the tile whose Sᵀ is computed next: kf[2ks + tt]
  half4_t vlo[16], vhi[16];   // Vᵀ fragments of the current tile: index 4g + dt, g = 2tt + u
  f32x16_t sx[4], sy[4];      // Sᵀ blocks b = 2qb + tt: one buffer holds E(t) / P(t), the other receives Sᵀ(t+1)
  half8_t pf[2][4];           // P fragments [qb][g]
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
  uint32_t v_ad[4];           // per dt: LDS address of the transpose reads of the current tile

  auto read_k = [&](int i, const char* slot) {   // i = 2ks + tt
    const int tt = i & 1, ks = i >> 1;
    kf[i] = *(const half8_t*)(slot + ((k_rd ^ (ks * 32)) + tt * 8192));
  };
  auto read_v_half = [&](auto ic) {   // one transpose read: fragment (ic >> 1) = 4g + dt, half ic & 1
    constexpr int c = decltype(ic)::value, i = c >> 1, g = i >> 2, dt = i & 3;
    constexpr int ro = (32 * (g >> 1) + 16 * (g & 1)) * 256;
    if constexpr ((c & 1) == 0) vlo[i] = lds_tr16_asm<ro>(v_ad[dt]);
    else vhi[i] = lds_tr16_asm<ro + 8 * 256>(v_ad[dt]);
  };

  float alpha[2] = {1.f, 1.f};
  bool need_rescale = false;
  float mxp[2][4];   // partial row maxima
  // partial row max of block b (16 values -> mxp[qb][2tt], [2tt+1])
  auto max_block = [&](f32x16_t (&s)[4], int b, int half) {
    const int qb = b >> 1, tt = b & 1, r0 = 8 * half;
    const float m0 = fmaxf(fmaxf(s[b][r0], s[b][r0 + 1]), s[b][r0 + 2]);
    const float m1 = fmaxf(fmaxf(s[b][r0 + 3], s[b][r0 + 4]), s[b][r0 + 5]);
    mxp[qb][2 * tt + half] = fmaxf(fmaxf(m0, m1), fmaxf(s[b][r0 + 6], s[b][r0 + 7]));
    asm volatile("" : "+v"(mxp[qb][2 * tt + half]));
  };
  auto decide = [&]() {
    float mc[2];
    bool ok = true;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      float mx = fmaxf(fmaxf(mxp[qb][0], mxp[qb][1]), fmaxf(mxp[qb][2], mxp[qb][3]));
      mx = aw4_xhalf_max(mx);
      mc[qb] = fmaxf(m_run[qb], mx * sl2);
      ok = ok && (mc[qb] - m_run[qb] <= RESCALE_THR);
    }
    need_rescale = !__all(ok);
    if (need_rescale) {
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        alpha[qb] = __builtin_amdgcn_exp2f(m_run[qb] - mc[qb]);
        m_run[qb] = mc[qb];
        l_run[qb] *= alpha[qb];
      }
    }
  };
  // E = S*scale - m for 4 values of block b (in place)
  auto scale4 = [&](f32x16_t (&s)[4], int b, int r0) {
    const int qb = b >> 1;
#pragma unroll
    for (int r = r0; r < r0 + 4; ++r) s[b][r] = __builtin_fmaf(s[b][r], sl2, -m_run[qb]);
    asm volatile("" : "+v"(s[b][r0]), "+v"(s[b][r0 + 1]), "+v"(s[b][r0 + 2]), "+v"(s[b][r0 + 3]));
  };
  auto rescale_o = [&]() {   // rare: Oᵀ *= alpha (all PV MFMAs issued; drain them first)
    aw4_drain();
    static_for<128>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      aw4_acc_scale<r>(alpha[r >> 6]);
    });
  };

  // ---- prologue compute: Sᵀ(0) -> sx, its row max / E(0), K(1) fragments
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  raw_barrier();
#pragma unroll
  for (int i = 0; i < 16; ++i) read_k(i, smem);
  static_for<32>([&](auto cc) {
    constexpr int c = decltype(cc)::value, ks = c >> 2, qb = (c >> 1) & 1, tt = c & 1;
    aw4_qk<192 + 4 * (8 * qb + ks), ks == 0>(sx[2 * qb + tt], kf[2 * ks + tt]);
  });
  aw4_drain();
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    max_block(sx, b, 0);
    max_block(sx, b, 1);
  }
  decide();          // m_run = -inf -> alpha = 0: O and l are still zero, nothing to scale
  need_rescale = false;
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int r0 = 0; r0 < 16; r0 += 4) scale4(sx, b, r0);
  if (T > 1) {   // first half (k-steps 0..3) of the K(1) fragments; the second half is read during phase 1
#pragma unroll
    for (int i = 0; i < 8; ++i) read_k(i, smem + AW4_SLOT);
  }

  unsigned long long* stamp = reinterpret_cast<unsigned long long*>(const_cast<half_t*>(Q));
  const bool stamping = STAMPS && blockIdx.x == 0 && wave == 0 && lane == 0;
  auto STAMP = [&](int t, int k) {
    if constexpr (STAMPS) {
      if (t >= 16 && t < 20) {
        const unsigned long long c = __builtin_readcyclecounter();
        if (stamping) stamp[(t - 16) * 4 + k] = c;
      }
    }
  };

  // one KV tile: sa holds E(t) (consumed), sb receives Sᵀ(t+1) and leaves as E(t+1)
  // (LAST = compile-time "no tile t+1": a run-time test per MFMA chunk costs a branch per MFMA, measured ~25 cycles)
  auto iteration = [&](auto lastc, int t, f32x16_t (&sa)[4], f32x16_t (&sb)[4]) {
    constexpr bool has_next = !decltype(lastc)::value;
    const char* cur = smem + (t & 3) * AW4_SLOT;
    const char* k1 = smem + ((t + 1) & 3) * AW4_SLOT;
    const char* k2 = smem + ((t + 2) & 3) * AW4_SLOT;
    // tile t+3 replaces the dead tile t-1; past the end the last tile is staged again (never read): no branch
    const int t3 = t + 3 < T ? t + 3 : T - 1;
    STAMP(t, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    raw_barrier();
    STAMP(t, 1);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) v_ad[dt] = lds_addr32(cur + v_rd + ((dt ^ v_sw) << 6));

    // =========================== phase 1: Sᵀ(t+1) MFMAs | softmax(t) exp / sums / pack | Vᵀ(t) reads | DMA(t+3)
    float ps[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    float pp0 = 0.f, pp1 = 0.f;   // the P pair exponentiated one MFMA ago (its add / pack ride behind the next MFMA)
    static_for<33>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      if constexpr (c < 32) {
        constexpr int ks = c >> 2, qb = (c >> 1) & 1, tt = c & 1;
        aw4_qk<192 + 4 * (8 * qb + ks), ks == 0>(sb[2 * qb + tt], kf[2 * ks + tt]);
        __builtin_amdgcn_sched_barrier(0);
      }
      // order behind the MFMA: plain VALU of the previous P pair, the LDS reads, and only then the two v_exp of this
      // pair (see AW4_POSTQN: an accumulating VGPR-SrcC MFMA needs a few instructions of distance to what follows)
      if constexpr (c > 0) {    // row sums + fp16 pack of pair c-1
        constexpr int d = c - 1, f = d >> 2, g = f >> 1, pq = f & 1, j = 2 * (d & 3);
        ps[pq][0] += pp0;
        ps[pq][1] += pp1;
        half2_t h = {(half_t)pp0, (half_t)pp1};
        asm volatile("" : "+v"(h), "+v"(ps[pq][0]), "+v"(ps[pq][1]));
        pf[pq][g][j] = h[0];
        pf[pq][g][j + 1] = h[1];
      }
      if constexpr (c < 32) {
        // loads BEHIND the MFMA: hipcc guards the MFMA's K fragment with a counted lgkmcnt that also covers the
        // (uncounted) asm transpose reads — ahead of the MFMA they would expose an LDS round trip per chunk
        read_v_half(cc);          // the 32 transpose reads of Vᵀ(t), one per MFMA
        if constexpr (c < 16 && (c & 1) == 0) read_k(8 + (c >> 1), k1);   // K(t+1) fragments of k-steps 4..7
        if constexpr ((c & 3) == 3) {
          issue_piece(c >> 2, t3, t + 3);
        }
        __builtin_amdgcn_sched_barrier(0);
        // exps of P pair c: fragment f = c>>2 (g = f>>1, query block f&1), values j, j+1
        constexpr int f = c >> 2, g = f >> 1, pq = f & 1, ptt = g >> 1, u = g & 1, j = 2 * (c & 3);
        pp0 = __builtin_amdgcn_exp2f(sa[2 * pq + ptt][8 * u + j]);
        pp1 = __builtin_amdgcn_exp2f(sa[2 * pq + ptt][8 * u + j + 1]);
        asm volatile("" : "+v"(pp0), "+v"(pp1));   // issued HERE
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    l_run[0] += ps[0][0] + ps[0][1];
    l_run[1] += ps[1][0] + ps[1][1];
    STAMP(t, 2);

    // =========================== phase 2: Oᵀ += Vᵀ(t)·Pᵀ(t) MFMAs | row max of Sᵀ(t+1), decision, E(t+1) | K(t+2) reads
    lds_tr16_wait16(vlo);
    lds_tr16_wait16(vhi);
    static_for<32>([&](auto cc) {
      constexpr int c = decltype(cc)::value, g = c >> 3, dt = (c >> 1) & 3, qb = c & 1;
      aw4_pv<16 * (4 * qb + dt)>(cat4(vlo[4 * g + dt], vhi[4 * g + dt]), pf[qb][g]);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr ((c & 3) == 1) read_k(c >> 2, k2);   // K(t+2) fragments of k-steps 0..3, one per 4 MFMAs
      if constexpr (has_next) {
        if constexpr (c < 8) {             // partial maxima, blocks in the order they were completed by phase 1
          max_block(sb, c >> 1, c & 1);
        } else if constexpr (c == 8) {
          decide();
        } else if constexpr (c >= 9 && c < 25) {   // 16 groups of 4 values
          scale4(sb, (c - 9) >> 2, ((c - 9) & 3) * 4);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    STAMP(t, 3);
    if constexpr (has_next) {
      if (need_rescale) {
        rescale_o();
        need_rescale = false;
      }
    }
  };
  using NotLast = std::integral_constant<bool, false>;
  using Last = std::integral_constant<bool, true>;
  int t = 0;
  for (; t + 2 < T; t += 2) {
    iteration(NotLast{}, t, sx, sy);
    iteration(NotLast{}, t + 1, sy, sx);
  }
  if (t + 1 < T) {      // two tiles left
    iteration(NotLast{}, t, sx, sy);
    iteration(Last{}, t + 1, sy, sx);
  } else {              // one tile left
    iteration(Last{}, t, sx, sy);
  }

  // ---- epilogue: O = Oᵀ / l
  aw4_drain();
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) l_run[qb] = 1.0f / aw4_xhalf_sum(l_run[qb]);
  static_for<2>([&](auto qc) {
    constexpr int qb = decltype(qc)::value;
    half_t* orow = Ob + (size_t)(q0 + 32 * qb + l32) * D;
    static_for<16>([&](auto ec) {
      constexpr int dt = decltype(ec)::value >> 2, rq = decltype(ec)::value & 3;
      constexpr int base = 16 * (4 * qb + dt) + 4 * rq;
      half4_t h;
      h[0] = (half_t)(aw4_acc_read<base + 0>() * l_run[qb]);
      h[1] = (half_t)(aw4_acc_read<base + 1>() * l_run[qb]);
      h[2] = (half_t)(aw4_acc_read<base + 2>() * l_run[qb]);
      h[3] = (half_t)(aw4_acc_read<base + 3>() * l_run[qb]);
      *(half4_t*)(orow + 32 * dt + 8 * rq + 4 * hi) = h;
    });
  });
}

}  // namespace lc
