// attn_fwd.hip — FlashAttention-2 forward for gfx950 (fp16 in/out, fp32 softmax, fp32 MFMA accumulate).
//
// Replaces (from-scratch CDNA4 design) the reference's split-Q forward kernels:
//   kernels/flash-attn/mma/basic/flash_attn_mma_split_q.cu:55-699      (split_q)
//   kernels/flash-attn/mma/basic/flash_attn_mma_share_qkv.cu:46-769    (shared_qkv)
// Semantics (same as the reference): O = softmax(Q Kᵀ / sqrt(D)) V per (batch, head); non-causal;
// Q,K,V,O [B,H,N,D] contiguous (V optionally [B,H,D,N]); online softmax over KV tiles of 64.
//
// MI355X design, per workgroup of NW wave64 (NW*32 query rows), one wave = 32 query rows:
//   * "swapped" products: Sᵀ = K·Qᵀ and Oᵀ = Vᵀ·Pᵀ with v_mfma_f32_32x32x16_f16, so every lane owns ONE
//     query row (q = lane & 31): row max / row sum / rescale are lane-local, the only cross-lane op is
//     one exchange with lane ^ 32 (the reference needs width-4 shuffles per mma quad, utils.h:152-168).
//   * the C-layout of Sᵀ (lane half `hi` holds kv 4hi+{0..3}, 8+4hi+{0..3}, ...) is used directly as the
//     k-slot assignment of the Pᵀ operand; the Vᵀ operand is fetched with ds_read_b64_tr_b16 on the same
//     kv rows, so P never moves between lanes (MFMA contracts over (half, slot) pairs symmetrically).
//   * Q lives in registers for the whole kernel; K and V tiles are register-staged into a 2-slot LDS
//     ring (global loads for tile t+1 issued before the math of tile t, LDS writes after it), padded row
//     strides: K rows +16 B (conflict-free ds_read_b128), V rows so that stride % 256 == 64 (the 4
//     rows of a tr-read half-wave land on disjoint bank quarters). One barrier per KV tile.
#pragma once
#include "lc_common.h"

namespace lc {

constexpr int KVB = 64;  // kv rows per tile
constexpr float RESCALE_THR = 8.0f;  // log2 units

// workgroup barrier that also publishes this wave's LDS writes (ds_write -> lgkmcnt(0) -> s_barrier)
LC_DEVINL void pp_sync() { __syncthreads(); }

template <int D>
struct AttnCfg {
  static constexpr int CH = D / 8;                      // 16-byte chunks per row
  static constexpr int KSTRIDE = D * 2 + 16;            // bytes
  static constexpr int VSTRIDE = (D == 32) ? 64 : ((D * 2) % 256 == 0 ? D * 2 + 64 : (D == 64 ? 192 : 320));
  static constexpr int VT_STRIDE = KVB * 2 + 16;        // Vᵀ tile rows: [D][64 kv] (+16 B pad)
  static constexpr int KBYTES = KVB * KSTRIDE;
  static constexpr int VBYTES = KVB * VSTRIDE;
  static constexpr int VTBYTES = D * VT_STRIDE;
};

template <int D, bool VT>
constexpr int attn_lds_bytes() {
  return 2 * (AttnCfg<D>::KBYTES + (VT ? AttnCfg<D>::VTBYTES : AttnCfg<D>::VBYTES));
}

// ABL (ablation bits, perf diagnosis only — results are WRONG when non-zero): 1 = no v_exp, 2 = no P·V,
// 4 = no Q·Kᵀ, 8 = no global loads / LDS staging after the prologue, 16 = no per-tile barrier.
template <int D, int NW, bool VT, int ABL = 0>
__global__ __launch_bounds__(NW * 64, (NW >= 4 ? 2 : 1)) void attn_fwd_kernel(
    const half_t* __restrict__ Q, const half_t* __restrict__ K, const half_t* __restrict__ V,
    half_t* __restrict__ O, int N, int nqb /* query blocks per (b,h) */, float sl2 /* (1/sqrt(D))*log2(e) */) {
  using C = AttnCfg<D>;
  constexpr int NT = NW * 64;
  constexpr int DT = D / 32;   // 32-wide d tiles of Oᵀ
  constexpr int DS = D / 16;   // k-steps of the QKᵀ contraction
  constexpr int VB = VT ? C::VTBYTES : C::VBYTES;
  constexpr int SLOT = C::KBYTES + VB;
  constexpr int K_CHUNKS = KVB * C::CH;
  constexpr int V_CHUNKS = VT ? D * 8 : KVB * C::CH;
  constexpr int KL = (K_CHUNKS + NT - 1) / NT;
  constexpr int VL = (V_CHUNKS + NT - 1) / NT;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = wave_id();
  const int hi = lane >> 5;
  const int l32 = lane & 31;

  // XCD-aware placement: each XCD owns a contiguous range of (b,h) problems, so the nqb workgroups that
  // re-read one head's K/V run on the same XCD (same L2) back to back.
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const size_t bh = id / nqb;
  const int q0 = (id - (int)bh * nqb) * (NW * 32) + wave * 32;
  const half_t* Qb = Q + bh * (size_t)N * D;
  const half_t* Kb = K + bh * (size_t)N * D;
  const half_t* Vb = V + bh * (size_t)N * D;
  half_t* Ob = O + bh * (size_t)N * D;

  // ---- Q fragments (B operand of Sᵀ = K·Qᵀ): lane holds Q[q0 + l32][16*s + 8*hi .. +8]
  half8_t qf[DS];
#pragma unroll
  for (int s = 0; s < DS; ++s) {
    qf[s] = *(const half8_t*)(Qb + (size_t)(q0 + l32) * D + 16 * s + 8 * hi);
  }

  // ---- register staging of one K/V tile; per-lane global / LDS offsets are hoisted out of the KV loop
  // (the kernel is instruction-issue bound: no per-tile address arithmetic beyond one 64-bit add per load)
  u32x4_t kst[KL], vst[VL];
  const half_t* kg[KL];
  const half_t* vg[VL];
  int kw[KL], vw[VL];
#pragma unroll
  for (int j = 0; j < KL; ++j) {
    const int idx = tid + j * NT;
    kg[j] = Kb + (size_t)idx * 8;
    kw[j] = (idx / C::CH) * C::KSTRIDE + (idx % C::CH) * 16;
  }
#pragma unroll
  for (int j = 0; j < VL; ++j) {
    const int idx = tid + j * NT;
    if constexpr (!VT) {
      vg[j] = Vb + (size_t)idx * 8;
      vw[j] = C::KBYTES + (idx / C::CH) * C::VSTRIDE + (idx % C::CH) * 16;
    } else {
      vg[j] = Vb + (size_t)(idx >> 3) * N + (idx & 7) * 8;
      vw[j] = C::KBYTES + (idx >> 3) * C::VT_STRIDE + (idx & 7) * 16;
    }
  }
  const size_t kstep = (size_t)KVB * D, vstep = VT ? (size_t)KVB : (size_t)KVB * D;
  auto load_tile = [&](int t) {
#pragma unroll
    for (int j = 0; j < KL; ++j)
      if (K_CHUNKS % NT == 0 || tid + j * NT < K_CHUNKS) kst[j] = *(const u32x4_t*)(kg[j] + t * kstep);
#pragma unroll
    for (int j = 0; j < VL; ++j)
      if (V_CHUNKS % NT == 0 || tid + j * NT < V_CHUNKS) vst[j] = *(const u32x4_t*)(vg[j] + t * vstep);
  };
  auto store_tile = [&](char* slot) {
#pragma unroll
    for (int j = 0; j < KL; ++j)
      if (K_CHUNKS % NT == 0 || tid + j * NT < K_CHUNKS) *(u32x4_t*)(slot + kw[j]) = kst[j];
#pragma unroll
    for (int j = 0; j < VL; ++j)
      if (V_CHUNKS % NT == 0 || tid + j * NT < V_CHUNKS) *(u32x4_t*)(slot + vw[j]) = vst[j];
  };

  // ---- lane-dependent LDS read offsets
  const int k_rd = l32 * C::KSTRIDE + hi * 16;  // + t*32*KSTRIDE + s*32
  int v_rd;                                     // V: + (32t+16u)*VSTRIDE [+8*VSTRIDE] + dt*64
  if constexpr (!VT) {
    const int i = lane & 15, gi = (lane >> 4) & 1;
    v_rd = C::KBYTES + (4 * hi + (i >> 2)) * C::VSTRIDE + (16 * gi + 4 * (i & 3)) * 2;
  } else {
    v_rd = C::KBYTES + l32 * C::VT_STRIDE + (4 * hi) * 2;  // + dt*32*VT_STRIDE + (32t+16u)*2 [+16]
  }

  f32x16_t o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  // waves 4..7 are dispatched second and lose every issue arbitration to waves 0..3 (priority, then age):
  // one static s_setprio for the younger half evens the two halves out (measured: the older wave idled
  // 1200 of 4300 cycles per tile at the barrier waiting for its partner)
  if (NW == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1);
  const int T = N / KVB;
  load_tile(0);
  store_tile(smem);
  // Retire the Q loads HERE: otherwise hipcc carries "qf may still be in flight" into the loop and guards
  // every Q·Kᵀ MFMA with an in-order vmcnt(N) that also drains the K/V prefetch issued a few instructions
  // earlier (the whole HBM latency lands inside the MFMA phase of every tile).
#pragma unroll
  for (int s = 0; s < DS; ++s) asm volatile("" : "+v"(qf[s]));
  __syncthreads();

  // ABL & 32: phase time stamps (s_memtime) of waves 0 and 4 of workgroup 0, tiles 16..19, written as u64
  // over the first bytes of Q (already in registers by then; diagnosis only, clobbers the input!)
  unsigned long long* stamp = reinterpret_cast<unsigned long long*>(const_cast<half_t*>(Q));
  const bool stamping = (ABL & 32) && blockIdx.x == 0 && (wave == 0 || wave == 4) && lane == 0;
  auto STAMP = [&](int t, int k) {
    if constexpr (ABL & 32) {
      if (t >= 16 && t < 20) {
        const unsigned long long c = __builtin_readcyclecounter();
        if (stamping) stamp[((wave >> 2) * 4 + (t - 16)) * 8 + k] = c;
      }
    }
  };
  for (int t = 0; t < T; ++t) {
    char* cur = smem + ((ABL & 8) ? 0 : (t & 1)) * SLOT;
    STAMP(t, 0);
    const bool more = !(ABL & 8) && t + 1 < T;

    // ---- Sᵀ = K·Qᵀ : two 32x32 tiles (kv 0..31, 32..63), 2*DS MFMAs in groups of GQ with the next
    // group's K fragments (ds_read_b128) in flight behind the current group's MFMAs.
    f32x16_t s[2];
    if constexpr (ABL & 4) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[tt][r] = (float)qf[r & (DS - 1)][r & 7] * (float)(t + r);
    } else {
      constexpr int GQ = (DS % 4 == 0) ? 4 : 2;    // fragments per group
      constexpr int NGQ = 2 * DS / GQ;
      half8_t kf[2][GQ];
      auto load_k = [&](int g, half8_t (&dst)[GQ]) {
#pragma unroll
        for (int i = 0; i < GQ; ++i) {
          const int idx = g * GQ + i, tt = idx & 1, ks = idx >> 1;  // two independent accumulator chains
          dst[i] = *(const half8_t*)(cur + k_rd + tt * 32 * C::KSTRIDE + ks * 32);
        }
      };
      load_k(0, kf[0]);
#pragma unroll
      for (int g = 0; g < NGQ; ++g) {
        if (g + 1 < NGQ) load_k(g + 1, kf[(g + 1) & 1]);
#pragma unroll
        for (int i = 0; i < GQ; ++i) {
          const int idx = g * GQ + i, tt = idx & 1, ks = idx >> 1;  // two independent accumulator chains
          s[tt] = mfma32(kf[g & 1][i], qf[ks], ks == 0 ? (f32x16_t)0.f : s[tt]);
        }
        // the K/V global prefetch of tile t+1 is issued behind the first MFMA group: a VMEM issue holds the
        // wave for ~80 cycles, which now overlap MFMAs already queued on the matrix pipe
        __builtin_amdgcn_sched_barrier(0);
        if (g == 0 && more) load_tile(t + 1);
      }
    }
    __builtin_amdgcn_sched_barrier(0);

    STAMP(t, 1);
    // ---- online softmax (log2 domain): lane owns query row q = l32, kv columns split with lane^32
    float mt[8];
#pragma unroll
    for (int r = 0; r < 8; ++r)
      mt[r] = fmaxf(fmaxf(s[0][r], s[0][r + 8]), fmaxf(s[1][r], s[1][r + 8]));
    float mx = fmaxf(fmaxf(fmaxf(mt[0], mt[1]), fmaxf(mt[2], mt[3])),
                     fmaxf(fmaxf(mt[4], mt[5]), fmaxf(mt[6], mt[7])));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_cand = fmaxf(m_run, mx * sl2);
    // Deferred rescale: while no row's max grows by more than 2^RESCALE_THR keep the old reference max
    // (P is then bounded by 2^RESCALE_THR, exact in fp32 sums and far inside fp16 range) and skip the
    // O / l rescale entirely.  The decision covers ONLY this tile's P, which is exponentiated below, and
    // the previous tile's P·V is already accumulated -> everything at the old scale is scaled exactly once.
    if (!__all(m_cand - m_run <= RESCALE_THR)) {
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_cand);
      m_run = m_cand;
      l_run *= alpha;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
    }
    STAMP(t, 2);
    // packed fp32 math (v_pk_fma_f32 / v_pk_add_f32): two values per scale-subtract and per row-sum add
    f32x2_t ps2[2] = {f32x2_t{0.f, 0.f}, f32x2_t{0.f, 0.f}};
    const f32x2_t sl2v = {sl2, sl2}, nm = {-m_run, -m_run};
    half8_t pf[2][2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          const f32x2_t sv = {s[tt][8 * u + j], s[tt][8 * u + j + 1]};
          const f32x2_t e = __builtin_elementwise_fma(sv, sl2v, nm);
          f32x2_t p;
          p[0] = (ABL & 1) ? e[0] : __builtin_amdgcn_exp2f(e[0]);
          p[1] = (ABL & 1) ? e[1] : __builtin_amdgcn_exp2f(e[1]);
          ps2[(j >> 1) & 1] += p;
          pf[tt][u][j] = (half_t)p[0];
          pf[tt][u][j + 1] = (half_t)p[1];
        }
      }
    }
    {
      const f32x2_t t2 = ps2[0] + ps2[1];
      l_run += t2[0] + t2[1];
    }

    // ---- Oᵀ += Vᵀ·Pᵀ : 4 (tt,u) groups of DT MFMAs on independent accumulators; the next group's Vᵀ
    // fragments (2 transpose reads each) are in flight behind the current group's MFMAs.
    if constexpr (ABL & 2) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int u = 0; u < 2; ++u) asm volatile("" ::"v"(pf[tt][u]));
    } else {
      half8_t vf[2][DT];
      auto load_v = [&](int g, half8_t (&dst)[DT]) {
        const int tt = g >> 1, u = g & 1;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          if constexpr (!VT) {
            const char* p = cur + v_rd + (32 * tt + 16 * u) * C::VSTRIDE + dt * 64;
            dst[dt] = cat4(lds_tr16(p), lds_tr16(p + 8 * C::VSTRIDE));
          } else {
            const char* p = cur + v_rd + dt * 32 * C::VT_STRIDE + (32 * tt + 16 * u) * 2;
            dst[dt] = cat4(*(const half4_t*)p, *(const half4_t*)(p + 16));
          }
        }
      };
      load_v(0, vf[0]);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (g < 3) load_v(g + 1, vf[(g + 1) & 1]);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] = mfma32(vf[g & 1][dt], pf[g >> 1][g & 1], o[dt]);
        if (g == 1) {   // LDS writes of tile t+1 ride behind the P·V MFMAs (the other ring slot is idle)
          __builtin_amdgcn_sched_barrier(0);
          if (more) store_tile(smem + ((t & 1) ^ 1) * SLOT);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }

    STAMP(t, 3);
    if ((ABL & 2) && more) store_tile(smem + ((t & 1) ^ 1) * SLOT);   // (P·V ablated: stage here)
    STAMP(t, 4);
    if (!(ABL & 16)) __syncthreads();
    STAMP(t, 5);
  }

  // ---- epilogue: O = Oᵀ / l ; lane holds row q, 4 consecutive d per register quad
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  half_t* orow = Ob + (size_t)(q0 + l32) * D;
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      half4_t h;
#pragma unroll
      for (int j = 0; j < 4; ++j) h[j] = (half_t)(o[dt][4 * rq + j] * inv);
      *(half4_t*)(orow + 32 * dt + 8 * rq + 4 * hi) = h;
    }
  }
}


// (The round-1 four-cluster role-split kernel, attn_fwd_c4_kernel — 8 waves, K/V by LDS-DMA, 0.95-1.04 PFLOP/s at config 3 —
// was retired at the end of round 2: attn_w4u.hip keeps its K-tile swizzle (16-B chunk c of row r at slot
// c ^ (r & 15) on unpadded 256-B rows) and its LDS-DMA staging; git history has the kernel.)

// ------------------------------------------------------------------------------------------------
// Large head dims (D = 256, 512): the FFPA-style fine-grained tiling of the reference's
// flash_attn_mma_tiling_qkv.cu:75-797 re-thought for CDNA4.  Q, K and V are all streamed through LDS in
// 64-wide d slices (O(1) LDS in D: 18 + 9 + 12 KiB), Sᵀ is accumulated over the D/64 slices, and the
// whole Oᵀ tile of a wave (32 rows x D, D/2 fp32 registers: 256 at D = 512) stays in registers — which is
// why this kernel runs ONE wave per SIMD with the full 512-entry VGPR/AGPR file (the reference keeps O in
// fp16 registers at d >= 256, tiling_qkv.cu:830-840; here it stays fp32).  Softmax identical to the
// kernels above.  Correctness-first: single-buffered slices, two barriers per slice.
// DO = output columns per workgroup: D = 512 is computed as two workgroups of 256 output columns each
// (both accumulate the full-D Sᵀ: 1.5x the MFMA work, but Oᵀ fits the register file without spills).
// The D/64 Q·Kᵀ slices and DO/64 P·V slices of a KV tile form one stream of stages; stage i+1's global
// loads are in flight (registers) while stage i computes from the 2-slot LDS ring: one barrier per stage.
template <int D, int DO, int NW, bool VT, bool BF16 = false>
__global__ __launch_bounds__(NW * 64, 1) void attn_fwd_bigd_kernel(
    const half_t* __restrict__ Q, const half_t* __restrict__ K, const half_t* __restrict__ V,
    half_t* __restrict__ O, int N, int nqb, float sl2) {
  constexpr int NT = NW * 64;
  constexpr int SL = 64;                 // d slice
  constexpr int NS = D / SL;             // slices of the Q·Kᵀ contraction
  constexpr int NSO = DO / SL;           // V / O slices owned by this workgroup
  constexpr int NSPLIT = D / DO;
  constexpr int QR = NW * 32;            // query rows per workgroup
  constexpr int QSTR = SL * 2 + 16;      // 144 B rows: conflict-free b128 fragment reads
  constexpr int VSTR = 192;              // V slice rows [64 kv][64 d]: stride % 256 == 192 (tr-read quarters)
  constexpr int Q_OFF = 0, K_OFF = QR * QSTR, V_OFF = 0;      // the V slice aliases the Q/K area of a slot
  constexpr int BUF = QR * QSTR + KVB * QSTR;                // one ring slot
  constexpr int QL = QR * 8 / NT, KL = KVB * 8 / NT;         // 16-byte chunks per thread: Q 4, K/V 2 (NW = 4)
  static_assert(QR * 8 % NT == 0 && KVB * 8 % NT == 0, "slice chunks must divide evenly");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = wave_id();
  const int hi = lane >> 5;
  const int l32 = lane & 31;

  const int id0 = xcd_remap(blockIdx.x, gridDim.x);
  const int dbase = (id0 % NSPLIT) * DO;   // first output column of this workgroup
  const int id = id0 / NSPLIT;
  const size_t bh = id / nqb;
  const int qblk = (id - (int)bh * nqb) * QR;
  const half_t* Qb = Q + bh * (size_t)N * D + (size_t)qblk * D;
  const half_t* Kb = K + bh * (size_t)N * D;
  const half_t* Vb = V + bh * (size_t)N * D;
  half_t* Ob = O + bh * (size_t)N * D;

  const int q_rd = Q_OFF + (wave * 32 + l32) * QSTR + hi * 16;   // + ks*32
  const int k_rd = K_OFF + l32 * QSTR + hi * 16;                 // + tt*32*QSTR + ks*32
  int v_rd;
  if constexpr (!VT) {
    const int i = lane & 15, gi = (lane >> 4) & 1;
    v_rd = V_OFF + (4 * hi + (i >> 2)) * VSTR + (16 * gi + 4 * (i & 3)) * 2;  // + (32tt+16u)*VSTR [+8*VSTR] + dt*64
  } else {
    v_rd = V_OFF + l32 * QSTR + (4 * hi) * 2;                                // + dt*32*QSTR + (32tt+16u)*2 [+16]
  }

  // ---- register staging of the next stage (Q slice + K slice, or V slice)
  u32x4_t rq[QL], rk[KL];
  auto load_qk = [&](int t, int sl) {
#pragma unroll
    for (int j = 0; j < QL; ++j) {
      const int idx = tid + j * NT, row = idx >> 3, c = idx & 7;
      rq[j] = *(const u32x4_t*)(Qb + (size_t)row * D + sl * SL + c * 8);
    }
#pragma unroll
    for (int j = 0; j < KL; ++j) {
      const int idx = tid + j * NT, row = idx >> 3, c = idx & 7;
      rk[j] = *(const u32x4_t*)(Kb + (size_t)(t * KVB + row) * D + sl * SL + c * 8);
    }
  };
  auto commit_qk = [&](char* buf) {
#pragma unroll
    for (int j = 0; j < QL; ++j) {
      const int idx = tid + j * NT, row = idx >> 3, c = idx & 7;
      *(u32x4_t*)(buf + Q_OFF + row * QSTR + c * 16) = rq[j];
    }
#pragma unroll
    for (int j = 0; j < KL; ++j) {
      const int idx = tid + j * NT, row = idx >> 3, c = idx & 7;
      *(u32x4_t*)(buf + K_OFF + row * QSTR + c * 16) = rk[j];
    }
  };
  auto load_v = [&](int t, int sl) {
#pragma unroll
    for (int j = 0; j < KL; ++j) {
      const int idx = tid + j * NT, row = idx >> 3, c = idx & 7;
      if constexpr (!VT)   // row = kv, c = d chunk
        rk[j] = *(const u32x4_t*)(Vb + (size_t)(t * KVB + row) * D + dbase + sl * SL + c * 8);
      else                 // row = d (64 of this slice), c = kv chunk
        rk[j] = *(const u32x4_t*)(Vb + (size_t)(dbase + sl * SL + row) * N + (size_t)t * KVB + c * 8);
    }
  };
  auto commit_v = [&](char* buf) {
#pragma unroll
    for (int j = 0; j < KL; ++j) {
      const int idx = tid + j * NT, row = idx >> 3, c = idx & 7;
      *(u32x4_t*)(buf + V_OFF + row * (VT ? QSTR : VSTR) + c * 16) = rk[j];
    }
  };

  f32x16_t o[DO / 32];
#pragma unroll
  for (int dt = 0; dt < DO / 32; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int T = N / KVB;
  int cur = 0;
  load_qk(0, 0);
  commit_qk(smem);
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    f32x16_t s[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[tt][r] = 0.f;
    // ---- Sᵀ += K[:, slice]·Q[:, slice]ᵀ over the D/64 slices
#pragma unroll 1
    for (int sl = 0; sl < NS; ++sl) {
      const bool last = sl + 1 == NS;
      if (!last) load_qk(t, sl + 1); else load_v(t, 0);
      const char* buf = smem + cur * BUF;
#pragma unroll
      for (int ks = 0; ks < SL / 16; ++ks) {
        const half8_t qfr = *(const half8_t*)(buf + q_rd + ks * 32);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const half8_t kfr = *(const half8_t*)(buf + k_rd + tt * 32 * QSTR + ks * 32);
          s[tt] = mfma32_16<BF16>(kfr, qfr, s[tt]);
        }
      }
      char* nxt = smem + (cur ^ 1) * BUF;
      if (!last) commit_qk(nxt); else commit_v(nxt);
      __syncthreads();
      cur ^= 1;
    }
    // ---- online softmax (same arithmetic as attn_fwd_kernel)
    float mt[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) mt[r] = fmaxf(fmaxf(s[0][r], s[0][r + 8]), fmaxf(s[1][r], s[1][r + 8]));
    float mx = fmaxf(fmaxf(fmaxf(mt[0], mt[1]), fmaxf(mt[2], mt[3])),
                     fmaxf(fmaxf(mt[4], mt[5]), fmaxf(mt[6], mt[7])));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_cand = fmaxf(m_run, mx * sl2);
    if (!__all(m_cand - m_run <= RESCALE_THR)) {
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_cand);
      m_run = m_cand;
      l_run *= alpha;
#pragma unroll
      for (int dt = 0; dt < DO / 32; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
    }
    float ps[4] = {0.f, 0.f, 0.f, 0.f};
    half8_t pf[2][2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[tt][8 * u + j], sl2, -m_run));
          ps[j & 3] += p;
          pf[tt][u][j] = cvt16<BF16>(p);
        }
    l_run += (ps[0] + ps[1]) + (ps[2] + ps[3]);
    // ---- Oᵀ[slice] += V[:, slice]ᵀ·Pᵀ, one 64-wide d slice of V per stage (fully unrolled: o[] indices static)
#pragma unroll
    for (int sl = 0; sl < NSO; ++sl) {
      const bool last = sl + 1 == NSO;
      const bool more = t + 1 < T;
      if (!last) load_v(t, sl + 1); else if (more) load_qk(t + 1, 0);
      const char* buf = smem + cur * BUF;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int tt = g >> 1, u = g & 1;
#pragma unroll
        for (int d2 = 0; d2 < 2; ++d2) {
          half8_t vfr;
          if constexpr (!VT) {
            const char* p = buf + v_rd + (32 * tt + 16 * u) * VSTR + d2 * 64;
            vfr = cat4(lds_tr16(p), lds_tr16(p + 8 * VSTR));
          } else {
            const char* p = buf + v_rd + d2 * 32 * QSTR + (32 * tt + 16 * u) * 2;
            vfr = cat4(*(const half4_t*)p, *(const half4_t*)(p + 16));
          }
          o[sl * 2 + d2] = mfma32_16<BF16>(vfr, pf[tt][u], o[sl * 2 + d2]);
        }
      }
      char* nxt = smem + (cur ^ 1) * BUF;
      if (!last) commit_v(nxt); else if (more) commit_qk(nxt);
      __syncthreads();
      cur ^= 1;
    }
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  half_t* orow = Ob + (size_t)(qblk + wave * 32 + l32) * D + dbase;
#pragma unroll
  for (int dt = 0; dt < DO / 32; ++dt) {
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      half4_t h;
#pragma unroll
      for (int j = 0; j < 4; ++j) h[j] = cvt16<BF16>(o[dt][4 * rq + j] * inv);
      *(half4_t*)(orow + 32 * dt + 8 * rq + 4 * hi) = h;
    }
  }
}

template <int NW>
constexpr int attn_bigd_lds_bytes() {
  return 2 * (NW * 32 * 144 + KVB * 144);   // two ring slots of (Q slice + K slice); V slices alias them
}

}  // namespace lc
