// attn_w4m.hip — FlashAttention-2 forward, D = 128: FOUR wave64 per workgroup, 64 query rows per wave, one wave per
// SIMD, ONE uniform "merged" phase per 32-row KV half-tile (round 2; successor of attn_w4.hip).
//
// Same semantics / entry points as attn_fwd.hip (reference: kernels/flash-attn/mma/basic/
// flash_attn_mma_split_q.cu:55-699, flash_attn_mma_share_qkv.cu:46-769).
//
// What round 1 measured: every schedule sat at ~1.0 PFLOP/s because a wave issued ~3.5 VALU per score
// element + 1.5 LDS reads per MFMA (65 issue cycles per 32-cycle MFMA), and attn_w4's Sᵀ-in-VGPR MFMAs needed an
// empirical 8-state pad each (the kernel spilled: hipcc moved S pieces right behind the asm MFMAs).  This kernel cuts
// the INSTRUCTION COUNT instead of trying another schedule:
//   * Q is pre-scaled by scale·log2e once (fp32 multiply, fp16 round) -> no per-element multiply;
//   * the running max is not subtracted per element either: the first Q·Kᵀ MFMA of a block takes C = −m (a 16-register
//     tuple holding the row's stale max), so the accumulator comes out as s − m directly;
//   * no per-element max: m is only a SCALE (any value keeps softmax exact as long as P fits fp16), so the common path
//     never recomputes it.  A tile whose row sums reach 2^14 (or are not finite) takes a rare wave-uniform slow path
//     that finds the true max, rescales O / l / the −m tuples and recomputes that tile's P (forced by the spike test);
//   * softmax per element = v_exp_f32 + v_add_f32 (row sum from the unrounded P, as split_q.cu:467-468) + half a
//     v_cvt_pk_f16_f32: 2.5 VALU per score element (round 1: 3.5-4.5);
//   * K fragments live in AGPRs (ds_read_b128 straight into a[128:191]), so the arch VGPRs hold only two 32-register Sᵀ
//     buffers, two P buffers, two 16-register Vᵀ sets and the −m tuples: no spills, nothing of hipcc's in the AGPRs
//     (leetcuda_amd/isa_audit.py checks that after every build);
//   * O leaves through LDS: whole 256-B rows with 16-B stores (the reference's "Os2g" idea, flash_attn.cc:217-219)
//     instead of 8-B stores at a 256-B stride.
// Pipeline: half-tile j = 32 KV rows.  Phase j issues 32 MFMAs, alternating
//       Sᵀ(j+1) = K(j+1)·Qᵀ (16)      and      Oᵀ += Vᵀ(j−1)·Pᵀ(j−1) (16)
// so consecutive MFMAs never touch the same accumulator block (distance 4 / 16), and carries in their issue shadow
//       softmax(j): 32 exp + 32 add + 16 cvt | LDS: 8 K reads (half-tile j+2) + 16 Vᵀ transpose reads | 8 k/v address adds
//       | DMA of tile t+2 (8 pieces, phase 0 of the tile only)
// ≈ 3.7 instructions per MFMA gap (budget of one wave per SIMD: 5, MI355X_MICROARCH.md "Per-instruction cycle constants").
// Register plan (literal AGPRs): a[0:127] Oᵀ(qb, dt) = a[16(4qb+dt)..]; a[128:191] two K fragment buffers
// (half-tile n uses buffer n & 1, fragment ks at +4ks); a[192:255] Q~ fragments Q(qb, ks) = a[192 + 4(8qb+ks)..].
// LDS: ring of 4 KV tiles of 64 rows (K 16 KiB + V 16 KiB, unpadded 256-B rows; K: 16-B chunk c of row r at slot c ^ (r & 15), V: 64-B unit u of row r at unit u ^ (r & 3)) filled by
// LDS-DMA; ONE barrier per 64-row tile.  Tile t+2 is staged during phase 2t (its slot held tile t−2, last read in phase
// 2t−2), every wave waits for its own pieces before the barrier that opens tile t+1's period, where K(t+2) is first read.
#pragma once
#include "attn_fwd.hip"

#define AM_COMMA ,

namespace lc {

constexpr int AM_TILE = KVB * 128 * 2;      // 16 KiB: one K or V tile (64 rows)
constexpr int AM_SLOT = 2 * AM_TILE;        // K + V
constexpr int AM_LDS = 4 * AM_SLOT;         // 128 KiB
constexpr int AM_EPI_STRIDE = 272;          // bytes per staged O row (256 + 16 pad)
constexpr int AM_O = 0, AM_K = 128, AM_Q = 192;
constexpr float AM_PSUM_LIMIT = 16384.0f;   // row-sum bound of one half-tile per lane: P <= 2^14 fits fp16 comfortably

// ---- asm statements on literal AGPRs.  All 256 AGPRs are asm-owned: every statement names them all as clobbers so
// hipcc never parks a value of its own there (audited).  hipcc pads no hazards around asm:
//   * K fragments reach the MFMA through an explicit s_waitcnt lgkmcnt(0) (am_lgkm0 / am_wait_v8);
//   * an S block is touched by every 4th MFMA only and read by VALU >= 2 MFMAs (>= 64 cycles) after its last write;
//   * P / V operands are written >= 16 MFMAs before the MFMA that reads them.
template <int R>
LC_DEVINL void am_acc_write(uint32_t x) { asm volatile("v_accvgpr_write_b32 a[%1], %0" :: "v"(x), "n"(R) : LC_AGPR_ALL); }
template <int R>
LC_DEVINL void am_acc_zero() { asm volatile("v_accvgpr_write_b32 a[%0], 0" :: "n"(R) : LC_AGPR_ALL); }
template <int R>
LC_DEVINL float am_acc_read() {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(x) : "n"(R) : LC_AGPR_ALL);
  return x;
}
template <int R>
LC_DEVINL void am_acc_scale(float alpha) {   // a[R] *= alpha (slow path; MFMAs drained by the caller)
  float tmp;
  asm volatile("v_accvgpr_read_b32 %0, a[%2]\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 1\n\tv_accvgpr_write_b32 a[%2], %0"
               : "=&v"(tmp) : "v"(alpha), "n"(R) : LC_AGPR_ALL);
}
LC_DEVINL void am_drain() { asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory"); }
// Drain in front of COMPILER-scheduled reads of MFMA results held in VGPRs: the registers must be operands.  A bare
// asm volatile is ordered against other volatile asm only — hipcc hoisted the v_max of the prologue's Sᵀ blocks above the
// wait states, right behind the MFMA that writes them (no interlock: the VALU read the accumulator one k-step short when
// issue was back to back and the full sum after an instruction-fetch stall -> a different but valid m, i.e. results that
// differed in the last bit between a cold and a warm launch).
LC_DEVINL void am_drain(f32x16_t& a, f32x16_t& b) {
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(a), "+v"(b)::"memory");
}
LC_DEVINL void am_drain(f32x16_t& a, f32x16_t& b, f32x16_t& c, f32x16_t& d) {
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory");
}
LC_DEVINL void am_drain(f32x4_t (&s)[2][4]) {
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15"
               : "+v"(s[0][0]), "+v"(s[0][1]), "+v"(s[0][2]), "+v"(s[0][3]), "+v"(s[1][0]), "+v"(s[1][1]), "+v"(s[1][2]), "+v"(s[1][3])
               :: "memory");
}
LC_DEVINL void am_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// K fragment: 16 bytes per lane straight into an AGPR quad
template <int AREG, int OFF>
LC_DEVINL void am_read_k(uint32_t lds_addr) {
  asm volatile("ds_read_b128 a[%1:%2], %0 offset:%3" :: "v"(lds_addr), "n"(AREG), "n"(AREG + 3), "n"(OFF) : LC_AGPR_ALL);
}
// ---- one MFMA slot = ONE asm statement: the MFMA followed by the LDS reads that ride in its issue shadow (hipcc pads a
// wait state at every asm-statement boundary that follows an asm load, so the reads live inside the MFMA's statement).
//   KIND 0: Sᵀ block = K fragment a[R0:+3] x Q~ fragment a[R1:+3] + C (first k-step; C = the −m tuple)
//   KIND 1: Sᵀ block += K fragment x Q~ fragment                      (accumulate in place)
//   KIND 2: Oᵀ block a[R0:+15] += Vᵀ fragment (VGPR) x Pᵀ fragment (VGPR)
//   KIND 3: no MFMA (slots of a phase whose product does not exist)
//   RD bit 0: + ds_read_b64_tr_b16 vout <- [vaddr + VOF];  bit 1: + ds_read_b128 a[KR:+3] <- [kaddr + KOF]
// PAD (A/B knob): wait states appended to an accumulating Q·Kᵀ MFMA.
#define AM_TXT_V "\n\tds_read_b64_tr_b16 %[vo], %[va] offset:%[vof]"
#define AM_TXT_K "\n\tds_read_b128 a[%[kr0]:%[kr1]], %[ka] offset:%[kof]"
#define AM_OPS_V [va] "v"(vaddr), [vof] "n"(VOF)
#define AM_OPS_K [ka] "v"(kaddr), [kr0] "n"(KR), [kr1] "n"(KR + 3), [kof] "n"(KOF)
#define AM_SLOT_BODY(MFMA_TXT, OUTS, INS)                                                                              \
  if constexpr (RD == 3)                                                                                               \
    asm volatile(MFMA_TXT AM_TXT_V AM_TXT_K : OUTS [vo] "=&v"(vout) : INS AM_OPS_V, AM_OPS_K : LC_AGPR_ALL);           \
  else if constexpr (RD == 1)                                                                                          \
    asm volatile(MFMA_TXT AM_TXT_V : OUTS [vo] "=&v"(vout) : INS AM_OPS_V : LC_AGPR_ALL);                              \
  else if constexpr (RD == 2)                                                                                          \
    asm volatile(MFMA_TXT AM_TXT_K : OUTS [dummy] "=&v"(vdummy) : INS AM_OPS_K : LC_AGPR_ALL);                         \
  else                                                                                                                 \
    asm volatile(MFMA_TXT : OUTS [dummy] "=&v"(vdummy) : INS [z] "n"(0) : LC_AGPR_ALL);
template <int KIND, int RD, int R0, int R1, int VOF, int KR, int KOF, int PAD = 0>
LC_DEVINL void am_slot(f32x16_t& sblk, const f32x16_t& cblk, half8_t vfrag, half8_t pfrag, half4_t& vout, uint32_t vaddr,
                       uint32_t kaddr) {
  uint32_t vdummy;   // keeps the operand lists uniform (an output is always present)
  if constexpr (KIND == 0) {
    AM_SLOT_BODY("v_mfma_f32_32x32x16_f16 %[s], a[%[r0]:%[r0e]], a[%[r1]:%[r1e]], %[c]", [s] "=&v"(sblk) AM_COMMA,
                 [c] "v"(cblk) AM_COMMA [r0] "n"(R0) AM_COMMA [r0e] "n"(R0 + 3) AM_COMMA [r1] "n"(R1) AM_COMMA [r1e] "n"(R1 + 3) AM_COMMA)
  } else if constexpr (KIND == 1 && PAD > 0) {
    AM_SLOT_BODY("v_mfma_f32_32x32x16_f16 %[s], a[%[r0]:%[r0e]], a[%[r1]:%[r1e]], %[s]\n\ts_nop %[pad]", [s] "+v"(sblk) AM_COMMA,
                 [pad] "n"(PAD - 1) AM_COMMA [r0] "n"(R0) AM_COMMA [r0e] "n"(R0 + 3) AM_COMMA [r1] "n"(R1) AM_COMMA [r1e] "n"(R1 + 3) AM_COMMA)
  } else if constexpr (KIND == 1) {
    AM_SLOT_BODY("v_mfma_f32_32x32x16_f16 %[s], a[%[r0]:%[r0e]], a[%[r1]:%[r1e]], %[s]", [s] "+v"(sblk) AM_COMMA,
                 [r0] "n"(R0) AM_COMMA [r0e] "n"(R0 + 3) AM_COMMA [r1] "n"(R1) AM_COMMA [r1e] "n"(R1 + 3) AM_COMMA)
  } else if constexpr (KIND == 2) {
    AM_SLOT_BODY("v_mfma_f32_32x32x16_f16 a[%[r0]:%[r0e]], %[vf], %[pf], a[%[r0]:%[r0e]]", ,
                 [vf] "v"(vfrag) AM_COMMA [pf] "v"(pfrag) AM_COMMA [r0] "n"(R0) AM_COMMA [r0e] "n"(R0 + 15) AM_COMMA)
  } else {
    AM_SLOT_BODY("", , )
  }
}
// (prologue / tail forms)
template <int KREG, int QREG>
LC_DEVINL void am_qk_zero(f32x16_t& s) {
  asm volatile("v_mfma_f32_32x32x16_f16 %0, a[%1:%2], a[%3:%4], 0"
               : "=&v"(s) : "n"(KREG), "n"(KREG + 3), "n"(QREG), "n"(QREG + 3) : LC_AGPR_ALL);
}
template <int KREG, int QREG>
LC_DEVINL void am_qk(f32x16_t& s) {
  asm volatile("v_mfma_f32_32x32x16_f16 %0, a[%1:%2], a[%3:%4], %0"
               : "+v"(s) : "n"(KREG), "n"(KREG + 3), "n"(QREG), "n"(QREG + 3) : LC_AGPR_ALL);
}
template <int OACC>
LC_DEVINL void am_pv(half8_t v, half8_t p) {
  asm volatile("v_mfma_f32_32x32x16_f16 a[%2:%3], %0, %1, a[%2:%3]"
               :: "v"(v), "v"(p), "n"(OACC), "n"(OACC + 15) : LC_AGPR_ALL);
}
LC_DEVINL void am_wait_v8(half4_t (&lo)[4], half4_t (&hi)[4]) {   // retire the asm transpose reads of one Vᵀ set
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]));
}
LC_DEVINL float am_xhalf_max(float x) {   // max over the two 32-lane halves (a row's kv columns are split with lane ^ 32)
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return fmaxf(a, b);
}
LC_DEVINL float am_xhalf_sum(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}

// PAD: wait states appended to every accumulating Q·Kᵀ MFMA (0 = none; the knob exists so that one GPU run can tell a
// hardware hazard from a codegen problem — round 1's attn_w4 needed 8 while it spilled)
template <int D, int PAD = 0>
__global__ __launch_bounds__(256) void attn_fwd_w4m_kernel(
    const half_t* __restrict__ Q, const half_t* __restrict__ K, const half_t* __restrict__ V,
    half_t* __restrict__ O, int N, int nqb, float sl2) {
  static_assert(D == 128, "w4m attention kernel: D = 128 only");
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
  const uint32_t smem32 = lds_addr32(smem);

  // ---- LDS-DMA: piece p = 4 rows x 256 B; this wave stages pieces wave + 4i (i = 0..3) of K and of V
  const int r4 = lane >> 4, cs = lane & 15;
  const unsigned k_off = (unsigned)(r4 * 256 + ((cs ^ (4 * wave + r4)) * 16));   // (row & 15) = 4(p&3) + r4, p&3 = wave
  const unsigned v_off = (unsigned)(r4 * 256 + ((cs ^ (r4 << 2)) * 16));         // (row & 3) = r4
  // buffer_load ... lds: descriptor (wave-uniform base of this (b,h)'s K / V) + scalar offset + one 32-bit lane offset —
  // 3 instructions per piece instead of the 64-bit per-lane address arithmetic of global_load_lds
  const buf_rsrc_t rk = make_rsrc(Kb), rv = make_rsrc(Vb);
  auto issue_piece = [&](int i, int t) {   // i = 0..7: K pieces, then V pieces; tile t (clamped) -> ring slot t & 3
    const int te = t < T ? t : T - 1;      // past the end: re-stage the last tile into a dead slot (never read)
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

  // ---- Q~ = fp16(Q * scale*log2e) -> AGPRs: lane holds Q[q0 + 32qb + l32][16 ks + 8 hi .. +8]
  static_for<16>([&](auto ic) {
    constexpr int i = decltype(ic)::value, qb = i >> 3, ks = i & 7;
    const half8_t q = *(const half8_t*)(Qb + (size_t)(q0 + 32 * qb + l32) * D + 16 * ks + 8 * hi);
    half8_t qs;
#pragma unroll
    for (int e = 0; e < 8; ++e) qs[e] = (half_t)((float)q[e] * sl2);
    const u32x4_t w = __builtin_bit_cast(u32x4_t, qs);
    am_acc_write<AM_Q + 4 * i + 0>(w[0]);
    am_acc_write<AM_Q + 4 * i + 1>(w[1]);
    am_acc_write<AM_Q + 4 * i + 2>(w[2]);
    am_acc_write<AM_Q + 4 * i + 3>(w[3]);
  });
  static_for<128>([&](auto r) { am_acc_zero<decltype(r)::value>(); });

  // ---- fragment read offsets inside a ring slot (swizzles: file header)
  uint32_t kx[8];   // K: row l32 (+32 per half-tile: immediate), 16-B chunk (2ks + hi) ^ (row & 15)
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) kx[ks] = (uint32_t)((l32 * 256 + ((hi ^ (l32 & 15)) * 16)) ^ (ks * 32));
  const int vi = lane & 15, vgi = (lane >> 4) & 1;
  uint32_t vx[4];   // Vᵀ transpose reads: kv row 4hi + (vi>>2) (+16g, +8: immediates), 64-B unit dt ^ (row & 3)
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
    vx[dt] = (uint32_t)(AM_TILE + (4 * hi + (vi >> 2)) * 256 + 32 * vgi + 8 * (vi & 3) + ((dt ^ (vi >> 2)) << 6));

  uint32_t ka[8], vc[4], vp[4];   // this tile period's LDS addresses: K(t+1) fragments, Vᵀ of tile t / tile t−1
  auto set_tile_addrs = [&](int t) {
    const uint32_t sb_cur = smem32 + (uint32_t)((t & 3) * AM_SLOT), sb_nxt = smem32 + (uint32_t)(((t + 1) & 3) * AM_SLOT);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      vp[dt] = vc[dt];
      vc[dt] = vx[dt] + sb_cur;
    }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) ka[ks] = kx[ks] + sb_nxt;
  };
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) vc[dt] = vx[dt] + smem32;

  f32x16_t sA[2], sB[2];        // Sᵀ blocks [qb] of the half-tile being exponentiated / being accumulated
  f32x16_t negm[2];             // C operand of the first k-step: 16 x (−m) per query block
  half8_t pA[2][2], pB[2][2];   // P fragments [qb][g] (g = 16-row k-step inside the half-tile)
  half4_t vlo0[4], vhi0[4];     // Vᵀ fragments, set 0: k-step g = 0 of the half-tile whose P·V runs next
  half4_t vlo1[4], vhi1[4];     // set 1: k-step g = 1
  float l_run[2] = {0.f, 0.f};

  // read one K half-tile (32 rows x 128) into AGPR buffer `BUF`: 8 fragments
  auto read_k_all = [&](auto bufc, uint32_t sbase, auto hc) {   // (prologue only)
    constexpr int BUF = decltype(bufc)::value, H = decltype(hc)::value;
    static_for<8>([&](auto kc) {
      constexpr int ks = decltype(kc)::value;
      am_read_k<AM_K + 32 * BUF + 4 * ks, H * 8192>(kx[ks] + sbase);
    });
  };
  // one transpose read of Vᵀ k-step G (0..3 inside its tile): fragment dt = c >> 1, half c & 1
  auto read_v = [&](auto cc, auto gc, const uint32_t (&va)[4], half4_t (&lo)[4], half4_t (&hv)[4]) {
    constexpr int c = decltype(cc)::value, G = decltype(gc)::value, dt = c >> 1;
    if constexpr ((c & 1) == 0) lo[dt] = lds_tr16_asm<G * 4096>(va[dt]);
    else hv[dt] = lds_tr16_asm<G * 4096 + 2048>(va[dt]);
  };

  // ---- prologue: tiles 0, 1 landed; K(0), K(1) -> AGPR buffers 0, 1; Sᵀ(0), its row max, E(0) = S − m, −m tuples
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  raw_barrier();
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  read_k_all(I0{}, smem32, I0{});
  read_k_all(I1{}, smem32, I1{});
  am_lgkm0();
  static_for<16>([&](auto ic) {
    constexpr int i = decltype(ic)::value, ks = i >> 1, qb = i & 1;
    if constexpr (ks == 0) am_qk_zero<AM_K + 4 * ks, AM_Q + 4 * (8 * qb + ks)>(sA[qb]);
    else am_qk<AM_K + 4 * ks, AM_Q + 4 * (8 * qb + ks)>(sA[qb]);
  });
  am_drain(sA[0], sA[1]);
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    float mx = sA[qb][0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sA[qb][r]);
    mx = am_xhalf_max(mx);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sA[qb][r] -= mx;
      negm[qb][r] = -mx;
    }
  }

  // ---- one merged phase.  H = half-tile parity (j = 2t + H); F: 1 = P·V(j−1) exists, 2 = Q·Kᵀ(j+1) exists,
  // 4 = K(j+2) exists (read it), 8 = issue the DMA of tile t+2.
  // sr = Sᵀ(j) (read), sw = Sᵀ(j+1) (written), pw = P(j) (written), pr = P(j−1) (read).
  auto phase = [&](auto hc, auto fc, int t, f32x16_t (&sr)[2], f32x16_t (&sw)[2], half8_t (&pw)[2][2],
                   half8_t (&pr)[2][2]) {
    constexpr int H = decltype(hc)::value, F = decltype(fc)::value;
    constexpr bool HAS_PV = (F & 1) != 0, HAS_QK = (F & 2) != 0, HAS_KRD = (F & 4) != 0, HAS_DMA = (F & 8) != 0;
    constexpr int KQ = AM_K + 32 * (1 - H);   // K(j+1) fragments: AGPR buffer (j+1) & 1
    constexpr int KRB = H;                     // K(j+2) goes to buffer (j+2) & 1 = H
    uint32_t (&v1a)[4] = H == 0 ? vp : vc;     // Vᵀ(j−1, g=1) lives in tile t−1 (H = 0) or tile t (H = 1)
    using G1 = std::integral_constant<int, H == 0 ? 3 : 1>;    // k-step of Vᵀ(j−1, g=1) inside its tile
    using G0 = std::integral_constant<int, 2 * H>;              // k-step of Vᵀ(j, g=0) inside tile t
    // set 0 (read in slots 16..23 of the previous phase) and K(j+1) are needed from slot 0 / 1 on
    am_wait_v8(vlo0, vhi0);
    float ps[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    float pp0 = 0.f, pp1 = 0.f, pq0 = 0.f, pq1 = 0.f;
    auto finish_pair = [&](auto pc, float e0, float e1) {
      constexpr int p = decltype(pc)::value, g = p >> 3, qb = (p >> 2) & 1, k = p & 3;
      ps[qb][0] += e0;
      ps[qb][1] += e1;
      half2_t h2 = {(half_t)e0, (half_t)e1};
      asm volatile("" : "+v"(h2), "+v"(ps[qb][0]), "+v"(ps[qb][1]));
      pw[qb][g][2 * k] = h2[0];
      pw[qb][g][2 * k + 1] = h2[1];
    };
    static_for<32>([&](auto sc) {
      constexpr int s = decltype(sc)::value, i = s >> 1;
      // ---------------- the MFMA of this slot + the LDS reads in its shadow, one asm statement (am_slot)
      // reads: slots 0..7: Vᵀ(j−1, g=1) -> set 1 (transpose read c = s) and K(j+2) fragment ks = s -> AGPR buffer KRB;
      //        slots 16..23: Vᵀ(j, g=0) -> set 0 (c = s − 16)
      constexpr bool RV1 = s < 8 && HAS_PV, RK = s < 8 && HAS_KRD, RV0 = s >= 16 && s < 24;
      constexpr int RD = ((RV1 || RV0) ? 1 : 0) | (RK ? 2 : 0);
      constexpr int c = RV0 ? s - 16 : (s & 7), rdt = c >> 1, rhalf = c & 1;
      constexpr int VOF = (RV0 ? 2 * H : (H == 0 ? 3 : 1)) * 4096 + rhalf * 2048;   // k-step inside its tile
      half4_t& vout = RV0 ? (rhalf ? vhi0[rdt] : vlo0[rdt]) : (rhalf ? vhi1[rdt] : vlo1[rdt]);
      const uint32_t vaddr = RV0 ? vc[rdt] : v1a[rdt];
      constexpr int KR = AM_K + 32 * KRB + 4 * (s & 7), KOF = H * 8192;
      if constexpr ((s & 1) == 0) {
        constexpr int ks = i >> 1, qb = i & 1;
        constexpr int KIND = HAS_QK ? (ks == 0 ? 0 : 1) : 3;
        if constexpr (KIND != 3 || RD != 0)
          am_slot<KIND, RD, KQ + 4 * ks, AM_Q + 4 * (8 * qb + ks), VOF, KR, KOF, PAD>(sw[qb], negm[qb], half8_t{}, half8_t{},
                                                                                    vout, vaddr, ka[s & 7]);
      } else {
        constexpr int g = i >> 3, dt = (i >> 1) & 3, qb = i & 1;
        constexpr int KIND = HAS_PV ? 2 : 3;
        if constexpr (KIND != 3 || RD != 0) {
          if constexpr (g == 0)
            am_slot<KIND, RD, AM_O + 16 * (4 * qb + dt), 0, VOF, KR, KOF>(sw[0], negm[0], cat4(vlo0[dt], vhi0[dt]),
                                                                       pr[qb][0], vout, vaddr, ka[s & 7]);
          else
            am_slot<KIND, RD, AM_O + 16 * (4 * qb + dt), 0, VOF, KR, KOF>(sw[0], negm[0], cat4(vlo1[dt], vhi1[dt]),
                                                                       pr[qb][1], vout, vaddr, ka[s & 7]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- compiler-scheduled fillers behind it
      if constexpr (s == 15 && HAS_PV) am_wait_v8(vlo1, vhi1);   // set 1 was read in slots 0..7; P·V g = 1 starts at slot 17
      if constexpr (HAS_DMA && (s & 3) == 3) issue_piece(s >> 2, t + 2);                  // 8 pieces, one per 4 slots
      // softmax(j): pair p -> k-step g = p >> 3, query block qb = (p >> 2) & 1, values 2k, 2k+1 of the 8.  Even slot 2p:
      // the two v_exp of pair p; odd slot 2p+1: row sums (from the unrounded P, split_q.cu:467-468) + fp16 pack of pair
      // p−1 — one pair LATER, so that hipcc sees other VALU between a v_exp and its consumer (gfx950 trans-use hazard:
      // otherwise it pads an s_nop in front of every consumer, 16 issue slots per phase).
      if constexpr ((s & 1) == 0) {
        constexpr int p = s >> 1, g = p >> 3, qb = (p >> 2) & 1, k = p & 3;
        pq0 = pp0;
        pq1 = pp1;
        pp0 = __builtin_amdgcn_exp2f(sr[qb][8 * g + 2 * k]);
        pp1 = __builtin_amdgcn_exp2f(sr[qb][8 * g + 2 * k + 1]);
        asm volatile("" : "+v"(pp0), "+v"(pp1));   // issued HERE
      } else if constexpr (s >= 3) {
        finish_pair(std::integral_constant<int, (s >> 1) - 1>{}, pq0, pq1);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    finish_pair(std::integral_constant<int, 15>{}, pp0, pp1);
    // ---------------- overflow guard: m is only a scale; redo this half-tile with the true max when P got large
    const float t0 = ps[0][0] + ps[0][1], t1 = ps[1][0] + ps[1][1];
    if (!__all(psum_below(t0, AM_PSUM_LIMIT) && psum_below(t1, AM_PSUM_LIMIT))) {   // (NaN / inf: bit-pattern compare, lc_common.h)
      am_drain(sw[0], sw[1]);                                  // every MFMA of this phase has written its result
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        float mx = sr[qb][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sr[qb][r]);
        mx = am_xhalf_max(mx);
        const float delta = fmaxf(mx, 0.f);                    // the row's max grew by `delta` (log2 units)
        const float alpha = __builtin_amdgcn_exp2f(-delta);
        l_run[qb] *= alpha;
        ps[qb][0] = 0.f;
        ps[qb][1] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          negm[qb][r] -= delta;
          if constexpr (HAS_QK) sw[qb][r] -= delta;             // Sᵀ(j+1) was accumulated against the old max
          const float pv = __builtin_amdgcn_exp2f(sr[qb][r] - delta);
          ps[qb][r & 1] += pv;
          pw[qb][r >> 3][r & 7] = (half_t)pv;
        }
        if (qb == 0) static_for<64>([&](auto rc) { am_acc_scale<AM_O + decltype(rc)::value>(alpha); });
        else static_for<64>([&](auto rc) { am_acc_scale<AM_O + 64 + decltype(rc)::value>(alpha); });
      }
      asm volatile("s_nop 3" ::: "memory");    // VALU writes of −m / S / P -> MFMA operand reads of the next phase
    }
    l_run[0] += ps[0][0] + ps[0][1];
    l_run[1] += ps[1][0] + ps[1][1];
  };
  using F_FIRST0 = std::integral_constant<int, 2 | 4 | 8>;       // j = 0: no P·V(−1)
  using F_MID = std::integral_constant<int, 1 | 2 | 4 | 8>;
  using F_MID1 = std::integral_constant<int, 1 | 2 | 4>;         // odd phases do not issue DMA
  using F_LAST0 = std::integral_constant<int, 1 | 2>;            // j = 2T−2: no tile T to read K from / to stage
  using F_LAST1 = std::integral_constant<int, 1>;                // j = 2T−1: no Q·Kᵀ(2T)

  // tile 0 (the prologue barrier already covers it)
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
    // tail: Oᵀ += Vᵀ(2T−1)·Pᵀ(2T−1); set 0 was read in the last phase, set 1 (k-step 3 of the last tile) now
    static_for<8>([&](auto cc) { read_v(cc, I3{}, vc, vlo1, vhi1); });
    am_wait_v8(vlo0, vhi0);
    am_wait_v8(vlo1, vhi1);
    static_for<16>([&](auto ic) {
      constexpr int i = decltype(ic)::value, g = i >> 3, dt = (i >> 1) & 3, qb = i & 1;
      if constexpr (g == 0) am_pv<AM_O + 16 * (4 * qb + dt)>(cat4(vlo0[dt], vhi0[dt]), pB[qb][0]);
      else am_pv<AM_O + 16 * (4 * qb + dt)>(cat4(vlo1[dt], vhi1[dt]), pB[qb][1]);
    });
  }

  // ---- epilogue: O = Oᵀ / l through LDS (whole 256-B rows, 16-B stores).  Lane holds O[q = 32qb + l32][d = 32dt + 8rq +
  // 4hi + (0..3)] in a[16(4qb+dt) + 4rq ..]; every wave owns a private 64 x 272 B staging area.
  am_drain();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  raw_barrier();                     // every wave is done with the KV ring
  float inv[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) inv[qb] = 1.0f / am_xhalf_sum(l_run[qb]);
  char* stg = smem + wave * (64 * AM_EPI_STRIDE);
  static_for<2>([&](auto qc) {
    constexpr int qb = decltype(qc)::value;
    static_for<16>([&](auto ec) {
      constexpr int dt = decltype(ec)::value >> 2, rq = decltype(ec)::value & 3;
      constexpr int base = AM_O + 16 * (4 * qb + dt) + 4 * rq;
      half4_t h;
      h[0] = (half_t)(am_acc_read<base + 0>() * inv[qb]);
      h[1] = (half_t)(am_acc_read<base + 1>() * inv[qb]);
      h[2] = (half_t)(am_acc_read<base + 2>() * inv[qb]);
      h[3] = (half_t)(am_acc_read<base + 3>() * inv[qb]);
      *(half4_t*)(stg + (32 * qb + l32) * AM_EPI_STRIDE + (32 * dt + 8 * rq + 4 * hi) * 2) = h;
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
