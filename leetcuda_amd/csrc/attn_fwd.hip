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

  // ---- register staging of one K/V tile
  u32x4_t kst[KL], vst[VL];
  auto load_tile = [&](int t) {
    const half_t* kp = Kb + (size_t)t * KVB * D;
#pragma unroll
    for (int j = 0; j < KL; ++j) {
      const int idx = tid + j * NT;
      if (K_CHUNKS % NT == 0 || idx < K_CHUNKS) kst[j] = *(const u32x4_t*)(kp + (size_t)idx * 8);
    }
#pragma unroll
    for (int j = 0; j < VL; ++j) {
      const int idx = tid + j * NT;
      if (V_CHUNKS % NT == 0 || idx < V_CHUNKS) {
        if constexpr (!VT) {
          vst[j] = *(const u32x4_t*)(Vb + (size_t)t * KVB * D + (size_t)idx * 8);
        } else {
          vst[j] = *(const u32x4_t*)(Vb + (size_t)(idx >> 3) * N + (size_t)t * KVB + (idx & 7) * 8);
        }
      }
    }
  };
  auto store_tile = [&](char* slot) {
#pragma unroll
    for (int j = 0; j < KL; ++j) {
      const int idx = tid + j * NT;
      if (K_CHUNKS % NT == 0 || idx < K_CHUNKS) {
        const int row = idx / C::CH, c = idx % C::CH;
        *(u32x4_t*)(slot + row * C::KSTRIDE + c * 16) = kst[j];
      }
    }
#pragma unroll
    for (int j = 0; j < VL; ++j) {
      const int idx = tid + j * NT;
      if (V_CHUNKS % NT == 0 || idx < V_CHUNKS) {
        if constexpr (!VT) {
          const int row = idx / C::CH, c = idx % C::CH;
          *(u32x4_t*)(slot + C::KBYTES + row * C::VSTRIDE + c * 16) = vst[j];
        } else {
          *(u32x4_t*)(slot + C::KBYTES + (idx >> 3) * C::VT_STRIDE + (idx & 7) * 16) = vst[j];
        }
      }
    }
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

  const int T = N / KVB;
  load_tile(0);
  store_tile(smem);
  // Retire the Q loads HERE: otherwise hipcc carries "qf may still be in flight" into the loop and guards
  // every Q·Kᵀ MFMA with an in-order vmcnt(N) that also drains the K/V prefetch issued a few instructions
  // earlier (the whole HBM latency lands inside the MFMA phase of every tile).
#pragma unroll
  for (int s = 0; s < DS; ++s) asm volatile("" : "+v"(qf[s]));
  __syncthreads();

  for (int t = 0; t < T; ++t) {
    char* cur = smem + ((ABL & 8) ? 0 : (t & 1)) * SLOT;
    if (!(ABL & 8) && t + 1 < T) load_tile(t + 1);

    // ---- Sᵀ = K·Qᵀ : two 32x32 tiles (kv 0..31, 32..63), 2*DS MFMAs in groups of GQ with the next
    // group's K fragments (ds_read_b128) in flight behind the current group's MFMAs.
    f32x16_t s[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[tt][r] = 0.f;
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
          s[tt] = mfma32(kf[g & 1][i], qf[ks], s[tt]);
        }
      }
      // pin the issue order: [GQ reads] then per group [GQ reads of the next group][GQ MFMAs]
      __builtin_amdgcn_sched_group_barrier(0x100, GQ, 0);
#pragma unroll
      for (int g = 0; g < NGQ; ++g) {
        if (g + 1 < NGQ) __builtin_amdgcn_sched_group_barrier(0x100, GQ, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, GQ, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);

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
    float ps[4] = {0.f, 0.f, 0.f, 0.f};
    half8_t pf[2][2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float e = __builtin_fmaf(s[tt][8 * u + j], sl2, -m_run);
          const float p = (ABL & 1) ? e : __builtin_amdgcn_exp2f(e);
          ps[j & 3] += p;
          pf[tt][u][j] = (half_t)p;
        }
      }
    }
    l_run += (ps[0] + ps[1]) + (ps[2] + ps[3]);

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
      }
    }

    if (!(ABL & 8) && t + 1 < T) store_tile(smem + ((t & 1) ^ 1) * SLOT);
    if (!(ABL & 16)) __syncthreads();
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


// ------------------------------------------------------------------------------------------------
// Ping-pong variant (8 waves, N % 256 == 0): the per-tile work of a wave is cut into
//   X(t) = Sᵀ = K·Qᵀ (16 MFMAs) + row max + (rare) rescale of O/l     — MFMA first, light VALU
//   Y(t) = P = exp2(...) / row sums / fp16 pack (VALU + transcendental) + Oᵀ += Vᵀ·Pᵀ (16 MFMAs)
// and the two waves that share a SIMD (wave w and w+4) run one barrier apart, so on every SIMD an X
// phase always faces a Y phase: the exp/VALU work of one wave hides behind the MFMAs of the other
// (in the lock-step kernel above both waves hit the matrix pipe and the VALU at the same moments).
// K(t+1) is written to the LDS ring at the end of X(t), V(t+1) at the end of Y(t); with group 1 one slot
// behind group 0 every write lands >= 1 barrier after the last read of the slot it replaces and >= 1
// barrier before its first read (K is only read in X phases, V only in Y phases).
template <int D, bool VT>
__global__ __launch_bounds__(512, 2) void attn_fwd_pp_kernel(
    const half_t* __restrict__ Q, const half_t* __restrict__ K, const half_t* __restrict__ V,
    half_t* __restrict__ O, int N, int nqb, float sl2) {
  using C = AttnCfg<D>;
  constexpr int NW = 8, NT = 512;
  constexpr int DT = D / 32, DS = D / 16;
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
  const int grp = wave >> 2;   // 0: leads, 1: one barrier behind (waves w and w+4 share a SIMD)
  const int hi = lane >> 5;
  const int l32 = lane & 31;

  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const size_t bh = id / nqb;
  const int q0 = (id - (int)bh * nqb) * (NW * 32) + wave * 32;
  const half_t* Qb = Q + bh * (size_t)N * D;
  const half_t* Kb = K + bh * (size_t)N * D;
  const half_t* Vb = V + bh * (size_t)N * D;
  half_t* Ob = O + bh * (size_t)N * D;

  half8_t qf[DS];
#pragma unroll
  for (int s = 0; s < DS; ++s) qf[s] = *(const half8_t*)(Qb + (size_t)(q0 + l32) * D + 16 * s + 8 * hi);

  u32x4_t kst[KL], vst[VL];
  auto load_k = [&](int t) {
#pragma unroll
    for (int j = 0; j < KL; ++j) {
      const int idx = tid + j * NT;
      if (K_CHUNKS % NT == 0 || idx < K_CHUNKS)
        kst[j] = *(const u32x4_t*)(Kb + (size_t)t * KVB * D + (size_t)idx * 8);
    }
  };
  auto load_v = [&](int t) {
#pragma unroll
    for (int j = 0; j < VL; ++j) {
      const int idx = tid + j * NT;
      if (V_CHUNKS % NT == 0 || idx < V_CHUNKS) {
        if constexpr (!VT)
          vst[j] = *(const u32x4_t*)(Vb + (size_t)t * KVB * D + (size_t)idx * 8);
        else
          vst[j] = *(const u32x4_t*)(Vb + (size_t)(idx >> 3) * N + (size_t)t * KVB + (idx & 7) * 8);
      }
    }
  };
  auto store_k = [&](char* slot) {
#pragma unroll
    for (int j = 0; j < KL; ++j) {
      const int idx = tid + j * NT;
      if (K_CHUNKS % NT == 0 || idx < K_CHUNKS)
        *(u32x4_t*)(slot + (idx / C::CH) * C::KSTRIDE + (idx % C::CH) * 16) = kst[j];
    }
  };
  auto store_v = [&](char* slot) {
#pragma unroll
    for (int j = 0; j < VL; ++j) {
      const int idx = tid + j * NT;
      if (V_CHUNKS % NT == 0 || idx < V_CHUNKS) {
        if constexpr (!VT)
          *(u32x4_t*)(slot + C::KBYTES + (idx / C::CH) * C::VSTRIDE + (idx % C::CH) * 16) = vst[j];
        else
          *(u32x4_t*)(slot + C::KBYTES + (idx >> 3) * C::VT_STRIDE + (idx & 7) * 16) = vst[j];
      }
    }
  };

  const int k_rd = l32 * C::KSTRIDE + hi * 16;
  int v_rd;
  if constexpr (!VT) {
    const int i = lane & 15, gi = (lane >> 4) & 1;
    v_rd = C::KBYTES + (4 * hi + (i >> 2)) * C::VSTRIDE + (16 * gi + 4 * (i & 3)) * 2;
  } else {
    v_rd = C::KBYTES + l32 * C::VT_STRIDE + (4 * hi) * 2;
  }

  f32x16_t o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  f32x16_t zero16;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero16[r] = 0.f;

  const int T = N / KVB;
  load_k(0);
  load_v(0);
  store_k(smem);
  store_v(smem);
#pragma unroll
  for (int s = 0; s < DS; ++s) asm volatile("" : "+v"(qf[s]));   // retire the Q loads before the loop
  pp_sync();
  if (grp == 1) pp_sync();

  for (int t = 0; t < T; ++t) {
    char* cur = smem + (t & 1) * SLOT;
    char* nxt = smem + ((t & 1) ^ 1) * SLOT;
    const bool more = t + 1 < T;
    // =========================== X(t) ===========================
    if (more) {
      load_k(t + 1);
      load_v(t + 1);
    }
    f32x16_t s[2];
    {
      constexpr int GQ = (DS % 4 == 0) ? 4 : 2;
      constexpr int NGQ = 2 * DS / GQ;
      half8_t kf[2][GQ];
      auto read_k = [&](int g, half8_t (&dst)[GQ]) {
#pragma unroll
        for (int i = 0; i < GQ; ++i) {
          const int idx = g * GQ + i, tt = idx & 1, ks = idx >> 1;  // two independent accumulator chains
          dst[i] = *(const half8_t*)(cur + k_rd + tt * 32 * C::KSTRIDE + ks * 32);
        }
      };
      read_k(0, kf[0]);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int g = 0; g < NGQ; ++g) {
        if (g + 1 < NGQ) read_k(g + 1, kf[(g + 1) & 1]);
#pragma unroll
        for (int i = 0; i < GQ; ++i) {
          const int idx = g * GQ + i, tt = idx & 1, ks = idx >> 1;  // two independent accumulator chains
          s[tt] = mfma32(kf[g & 1][i], qf[ks], ks == 0 ? zero16 : s[tt]);
        }
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_group_barrier(0x100, GQ, 0);
#pragma unroll
      for (int g = 0; g < NGQ; ++g) {
        if (g + 1 < NGQ) __builtin_amdgcn_sched_group_barrier(0x100, GQ, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, GQ, 0);
      }
    }
    {
      float mt[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) mt[r] = fmaxf(fmaxf(s[0][r], s[0][r + 8]), fmaxf(s[1][r], s[1][r + 8]));
      float mx = fmaxf(fmaxf(fmaxf(mt[0], mt[1]), fmaxf(mt[2], mt[3])),
                       fmaxf(fmaxf(mt[4], mt[5]), fmaxf(mt[6], mt[7])));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float m_cand = fmaxf(m_run, mx * sl2);
      if (!__all(m_cand - m_run <= RESCALE_THR)) {   // PV(t-1) is complete: O, l are all at the old scale
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_cand);
        m_run = m_cand;
        l_run *= alpha;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
      }
    }
    if (more) store_k(nxt);
    pp_sync();
    // =========================== Y(t) ===========================
    {
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
            pf[tt][u][j] = (half_t)p;
          }
      l_run += (ps[0] + ps[1]) + (ps[2] + ps[3]);
      half8_t vf[2][DT];
      auto read_v = [&](int g, half8_t (&dst)[DT]) {
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
      read_v(0, vf[0]);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (g < 3) read_v(g + 1, vf[(g + 1) & 1]);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] = mfma32(vf[g & 1][dt], pf[g >> 1][g & 1], o[dt]);
      }
    }
    if (more) store_v(nxt);
    pp_sync();
  }
  if (grp == 0) pp_sync();

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

}  // namespace lc
