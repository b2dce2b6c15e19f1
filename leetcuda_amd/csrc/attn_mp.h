// attn_mp.h — primitives shared by the merged-phase attention kernels (attn_w4u.hip, attn_w4i.hip) and the large-head-dim
// kernels (attn_bigd2.hip, attn_bigd4.hip): asm statements on LITERAL AGPRs, drains, the geometry of the 4-slot K / V ring.
// (Round 4: collected from the retired attn_w4m.hip / attn_w4n.hip / attn_w4g.hip — one copy of every statement.)
//
// Reference semantics served by these kernels: kernels/flash-attn/mma/basic/flash_attn_mma_split_q.cu:55-699 (dispatcher
// :769-815), flash_attn_mma_share_qkv.cu:46-769.
#pragma once
#include "attn_fwd.hip"

#define AM_COMMA ,

namespace lc {

constexpr float AM_PSUM_LIMIT = 16384.0f;   // row-sum bound of one half-tile per lane: P <= 2^14 fits fp16 comfortably

// introspection (lc_attn_slowpath_stats): how often the overflow slow path ran — [0] executions, [1] sum of half-tile indices
// j, [2] how many of them saw a non-finite row sum, [3] bit pattern of the last offending row sum.  One atomic per execution
// of a path that N(0,1) inputs never take.  One counter block per translation unit that instantiates these kernels:
// LC_AN_SLOWPATH_SYM names it.
#ifndef LC_AN_SLOWPATH_SYM
#define LC_AN_SLOWPATH_SYM g_an_slowpath
#endif
__device__ unsigned int LC_AN_SLOWPATH_SYM[4];

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
LC_DEVINL void am_drain(f32x16_t& a) { asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(a)::"memory"); }
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
LC_DEVINL void am_wait_v8(half4_t (&lo)[4], half4_t (&hi)[4]) {   // retire the asm Vᵀ reads of one set of four blocks
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]));
}
LC_DEVINL void w4g_wait_v4(half4_t& a, half4_t& b, half4_t& c, half4_t& d) {   // (am_wait_v8 for a set of two blocks)
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
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
// plain 8-byte LDS read as an asm statement (the V-transposed kernels' counterpart of lds_tr16_asm: hipcc would guard a
// builtin LDS load with vmcnt(0) after LDS-DMA just the same)
template <int OFF>
LC_DEVINL half4_t lds_rd64_asm(uint32_t lds_byte_addr) {
  half4_t r;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(lds_byte_addr), "n"(OFF));
  return r;
}

// ---- one MFMA slot = ONE asm statement: the MFMA followed by the LDS reads that ride in its issue shadow (hipcc pads a
// wait state at every asm-statement boundary that follows an asm load, so the reads live inside the MFMA's statement).
//   KIND 0: Sᵀ block = K fragment a[R0:+3] x Q~ fragment a[R1:+3] + C (first k-step; C = the −m tuple)
//   KIND 1: Sᵀ block += K fragment x Q~ fragment                      (accumulate in place)
//   KIND 2: Oᵀ block a[R0:+3] += Vᵀ fragment (VGPR) x Pᵀ fragment (VGPR)
//   KIND 3: no MFMA (slots of a phase whose product does not exist)
//   RD bit 0: + an 8-byte Vᵀ read vout <- [vaddr + VOF] — ds_read_b64_tr_b16 (V as [N][D] in LDS: the hardware 4 x 16
//             transpose) or, VT, a plain ds_read_b64 (V handed over as [D][N]: the rows already ARE Vᵀ);
//   RD bit 1: + ds_read_b128 a[KR:+3] <- [kaddr + KOF].
#define AM_TXT_VTR "\n\tds_read_b64_tr_b16 %[vo], %[va] offset:%[vof]"
#define AM_TXT_VPL "\n\tds_read_b64 %[vo], %[va] offset:%[vof]"
#define AM_TXT_K "\n\tds_read_b128 a[%[kr0]:%[kr1]], %[ka] offset:%[kof]"
#define AM_OPS_V [va] "v"(vaddr), [vof] "n"(VOF)
#define AM_OPS_K [ka] "v"(kaddr), [kr0] "n"(KR), [kr1] "n"(KR + 3), [kof] "n"(KOF)
// (OUTS / INS are parenthesised operand-list prefixes — "([s] "=&v"(x),)" — so that their commas survive the two macro levels)
#define AM_STRIP(...) __VA_ARGS__
#define AM_SLOT_BODY_V(MFMA_TXT, TXT_V, OUTS, INS)                                                                     \
  if constexpr (RD == 3)                                                                                               \
    asm volatile(MFMA_TXT TXT_V AM_TXT_K : AM_STRIP OUTS [vo] "=&v"(vout) : AM_STRIP INS AM_OPS_V, AM_OPS_K : LC_AGPR_ALL); \
  else if constexpr (RD == 1)                                                                                          \
    asm volatile(MFMA_TXT TXT_V : AM_STRIP OUTS [vo] "=&v"(vout) : AM_STRIP INS AM_OPS_V : LC_AGPR_ALL);               \
  else if constexpr (RD == 2)                                                                                          \
    asm volatile(MFMA_TXT AM_TXT_K : AM_STRIP OUTS [dummy] "=&v"(vdummy) : AM_STRIP INS AM_OPS_K : LC_AGPR_ALL);       \
  else                                                                                                                 \
    asm volatile(MFMA_TXT : AM_STRIP OUTS [dummy] "=&v"(vdummy) : AM_STRIP INS [z] "n"(0) : LC_AGPR_ALL);
#define AM_SLOT_BODY(MFMA_TXT, OUTS, INS)                                                                              \
  if constexpr (VT) { AM_SLOT_BODY_V(MFMA_TXT, AM_TXT_VPL, OUTS, INS) } else { AM_SLOT_BODY_V(MFMA_TXT, AM_TXT_VTR, OUTS, INS) }

template <int KIND, int RD, int R0, int R1, int VOF, int KR, int KOF, bool VT = false>
LC_DEVINL void an_slot(f32x4_t& sblk, const f32x4_t& cblk, half8_t vfrag, half8_t pfrag, half4_t& vout, uint32_t vaddr,
                       uint32_t kaddr) {
  uint32_t vdummy;   // keeps the operand lists uniform (an output is always present)
  if constexpr (KIND == 0) {
    AM_SLOT_BODY("v_mfma_f32_16x16x32_f16 %[s], a[%[r0]:%[r0e]], a[%[r1]:%[r1e]], %[c]", ([s] "=&v"(sblk),),
                 ([c] "v"(cblk), [r0] "n"(R0), [r0e] "n"(R0 + 3), [r1] "n"(R1), [r1e] "n"(R1 + 3),))
  } else if constexpr (KIND == 1) {
    AM_SLOT_BODY("v_mfma_f32_16x16x32_f16 %[s], a[%[r0]:%[r0e]], a[%[r1]:%[r1e]], %[s]", ([s] "+v"(sblk),),
                 ([r0] "n"(R0), [r0e] "n"(R0 + 3), [r1] "n"(R1), [r1e] "n"(R1 + 3),))
  } else if constexpr (KIND == 2) {
    AM_SLOT_BODY("v_mfma_f32_16x16x32_f16 a[%[r0]:%[r0e]], %[vf], %[pf], a[%[r0]:%[r0e]]", (),
                 ([vf] "v"(vfrag), [pf] "v"(pfrag), [r0] "n"(R0), [r0e] "n"(R0 + 3),))
  } else {
    AM_SLOT_BODY("", (), ())
  }
}
// (prologue / tail forms)
template <int KREG, int QREG>
LC_DEVINL void an_qk_zero(f32x4_t& s) {
  asm volatile("v_mfma_f32_16x16x32_f16 %0, a[%1:%2], a[%3:%4], 0"
               : "=&v"(s) : "n"(KREG), "n"(KREG + 3), "n"(QREG), "n"(QREG + 3) : LC_AGPR_ALL);
}
template <int KREG, int QREG>
LC_DEVINL void an_qk(f32x4_t& s) {
  asm volatile("v_mfma_f32_16x16x32_f16 %0, a[%1:%2], a[%3:%4], %0"
               : "+v"(s) : "n"(KREG), "n"(KREG + 3), "n"(QREG), "n"(QREG + 3) : LC_AGPR_ALL);
}
template <int OACC>
LC_DEVINL void an_pv(half8_t v, half8_t p) {
  asm volatile("v_mfma_f32_16x16x32_f16 a[%2:%3], %0, %1, a[%2:%3]"
               :: "v"(v), "v"(p), "n"(OACC), "n"(OACC + 3) : LC_AGPR_ALL);
}
// reductions over the four 16-lane groups (a query row's kv columns are spread over lanes l, l^16, l^32, l^48); only
// the prologue, the slow path and the epilogue use them
LC_DEVINL float an_x4_max(float x) {
  x = fmaxf(x, __shfl_xor(x, 16));
  return fmaxf(x, __shfl_xor(x, 32));
}
LC_DEVINL float an_x4_sum(float x) {
  x += __shfl_xor(x, 16);
  return x + __shfl_xor(x, 32);
}

// ---- geometry of the merged-phase kernels (4 waves x 64 query rows, v_mfma_f32_16x16x32_f16, ring of four 64-row K / V tiles):
//   NDS = D / 32  d-steps of Q·Kᵀ,  NDB = D / 16  column blocks of Oᵀ,
//   a phase (one 32-row KV half-tile, 64 query rows per wave) = 8 NDS Q·Kᵀ MFMAs alternating with 4 NDB = 8 NDS P·V MFMAs
//   = 16 NDS slots (64 at D = 128, 32 at D = 64) and ALWAYS 32 score elements per lane;
//   rows are 2 D bytes: K tile [64][2 D] with 16-B chunk c of row r at slot c ^ (r & 15) (D = 128) / c ^ ((r >> 1) & 7)
//   (D = 64: two rows share a 256-B bank row), V tile with 32-B column pair p of row r at pair slot p ^ key(r),
//   key = ((r & 3) << 1) | ((r >> 2) & 1) (D = 128) / (r >> 1) & 3 (D = 64) — conflict-free for the fragment reads of
//   both widths (tests/test_layouts.py); an LDS-DMA piece (1 KiB) = 4 rows (D = 128) / 8 rows (D = 64), 8 / 4 pieces per
//   wave and tile; ring of 4 tiles = 128 KiB (D = 128) / 64 KiB (D = 64).
//   V handed over TRANSPOSED ([B,H,D,N], the reference's *_swizzle_qkv entries): the V image of a tile is [D rows][64 kv] —
//   128-B rows whatever D is — with 16-B granule j of row d at slot j ^ ((d >> 1) & 7); a piece = 8 d-rows of 128 B
//   (source stride 2 N bytes); a fragment read = two plain ds_read_b64 (kv 4 g .. + 3 of kv block 0 and of kv block 1).
// AGPRs: a[0 : 16 NDB) Oᵀ blocks (db, qb) at 4 (4 db + qb); then two K half-tile buffers of 8 NDS registers (fragment
// (kvb, ds) at 4 (NDS kvb + ds)); then Q~ fragments (qb, ds) at 4 (NDS qb + ds) — 256 registers at D = 128, 128 at D = 64.
template <int D>
struct W4G {
  // ROWB = bytes per K / V row IN LDS; GROWB = in global memory.  D = 96 / 32 (attn_w4i.hip only) keep the 256-B / 128-B LDS rows of
  // D = 128 / 64 — 12 of 16 (4 of 8) chunks are real, the LDS-DMA lanes of the others re-fetch a valid chunk — so every layout formula
  // of the wider head dim holds.
  static constexpr int NDS = D / 32, NDB = D / 16, GROWB = 2 * D, ROWB = D > 64 ? 256 : 128;
  static constexpr int TILE = KVB * ROWB, SLOT = 2 * TILE, LDS = 4 * SLOT;
  static constexpr int NS = 16 * NDS;                 // MFMA slots per phase
  static constexpr int NRV = NDB, NRK = 2 * NDS;      // 8-byte reads per Vᵀ set (NDB / 2 blocks x 2), K fragments per half-tile
  static constexpr int RPP = 1024 / ROWB;             // rows per LDS-DMA piece
  static constexpr int PPW = TILE / 1024 / 4;         // pieces per wave and operand
  static constexpr int KBUF = 8 * NDS;                // AGPRs per K half-tile buffer
  static constexpr int O = 0, K = 16 * NDB, Q = K + 2 * KBUF;
  static constexpr int EPI_STRIDE = GROWB + 16;
  static_assert(D == 32 || D == 64 || D == 96 || D == 128, "merged-phase geometry: D = 32, 64, 96 or 128");
  static_assert(4 * 64 * EPI_STRIDE <= LDS, "epilogue staging must fit the ring");
};

}  // namespace lc
