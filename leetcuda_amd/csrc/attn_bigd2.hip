// attn_bigd2.hip — FlashAttention-2 forward for LARGE head dims (D = 256, 512; fp16 and bf16): one workgroup owns ALL D
// output columns of its 128 query rows (round 2; replaces the column-split attn_fwd_bigd_kernel for these shapes).
//
// Reference: the FFPA ancestor kernels/flash-attn/mma/basic/flash_attn_mma_tiling_qkv.cu:75-797 (fine-grained d tiling,
// O(1) SRAM in D), entry flash_attn_mma_stages_split_q_tiling_qkv (:881-945).  BASELINE config 5a: (1,48,8192,512).
//
// Why round 1's kernel stopped at 281 TFLOP/s (11 %): it split D = 512 into two workgroups of 256 output columns that
// BOTH recomputed the full-D Q·Kᵀ (1.5x the MFMA work), re-streamed the Q slices from global memory for every KV tile,
// and ran 12 register-staged 8-MFMA stages per tile with a __syncthreads() each.  Here:
//   * four wave64, one per SIMD with the whole 512-entry register file; a wave owns 32 query rows x all D columns:
//       Oᵀ accumulators  a[0 : D/2)   literal AGPRs (256 registers at D = 512), written only by asm P·V MFMAs
//       Q fragments      D/4 arch VGPRs (128 at D = 512), loaded ONCE — Q is never re-read
//       Sᵀ               4 blocks (2 KV halves x even/odd k-step partial sums: no MFMA depends on the previous three)
//     -> every MFMA issued is useful work (no recomputation), waves exchange nothing;
//   * KV tile = 64 rows: K tile 64 x D and V tile 64 x D live in LDS once (2 x 64 KiB at D = 512, single-buffered) and
//     are filled by LDS-DMA (buffer_load ... lds, one 1-KiB row per wave-instruction) in the shadow of the OTHER phase:
//         phase QK(t):  D/16 x 2 MFMAs  Sᵀ = K(t)·Qᵀ       | DMA of V(t)   (V region free since P·V(t−1))
//         softmax(t)    fp32, row sums from the unrounded P (tiling_qkv.cu keeps the same order)
//         barrier       (K(t) dead for everybody, V(t) landed)
//         phase PV(t):  D/32 x 4 MFMAs  Oᵀ += Vᵀ(t)·Pᵀ(t)   | DMA of K(t+1) (K region free since the barrier)
//         barrier       (V(t) dead, K(t+1) landed)
//     a DMA piece has >= 32 MFMAs (>= 1000 cycles) of flight; K/V bytes per MFMA are a third of round 1's;
//   * swizzles (LDS-DMA writes lane-linearly, so they are applied to the per-lane SOURCE address): K 16-B chunk c of row r
//     at slot c ^ (r & 15) (conflict-free ds_read_b128 on 1-KiB rows), V 64-B unit u of row r at unit u ^ (r & 3)
//     (transpose-read half-waves on disjoint bank quarters) — the layouts of the round-2 merged-phase kernel on longer rows;
//   * softmax: the running max is only a SCALE (as in attn_w4u.hip): the fast path exponentiates against the stale max and a
//     rare wave-uniform slow path (row sums >= 2^14, non-finite, or the first tile) finds the true max and rescales O, l;
//   * O leaves through LDS as whole rows with 16-B stores.
// Roofline: MFMA-bound (4·B·H·N²·D FLOPs, 4·B·H·N·D·2 algorithmic bytes; AI = N/2 FLOP/B per... >> 300).
#pragma once
#include "attn_fwd.hip"
#include "attn_mp.h"   // am_acc_* / am_drain / am_xhalf_* helpers

namespace lc {

// D = 512 runs at the 256-VGPR limit (Oᵀ fills the 256 AGPRs): the Q fragments of the last BD2_PARK k-steps live in the
// 32 KiB of LDS that the two 64-KiB tiles leave free (8 KiB per wave, lane-private 16-B slots) and come back through a
// three-deep register ring during the Q·Kᵀ phase — 32 VGPRs that the P·V + softmax phase needs.
template <int D>
constexpr int bigd2_park_ks() { return D == 512 ? 8 : 0; }
template <int D>
constexpr int bigd2_lds_bytes() {   // K tile + V tile (+ parked Q); the epilogue's O staging (4 waves x 32 rows x (2D + 16) B) aliases them
  constexpr int tiles = 2 * KVB * D * 2 + 4 * bigd2_park_ks<D>() * 1024, stage = 4 * 32 * (2 * D + 16);
  return tiles > stage ? tiles : stage;
}

// Oᵀ block a[R0:R0+15] += Vᵀ fragment x Pᵀ fragment (fp16 / bf16)
template <int R0, bool BF16>
LC_DEVINL void bd2_pv(half8_t v, half8_t p) {
  if constexpr (BF16)
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%2:%3], %0, %1, a[%2:%3]" :: "v"(v), "v"(p), "n"(R0), "n"(R0 + 15) : LC_AGPR_ALL);
  else
    asm volatile("v_mfma_f32_32x32x16_f16 a[%2:%3], %0, %1, a[%2:%3]" :: "v"(v), "v"(p), "n"(R0), "n"(R0 + 15) : LC_AGPR_ALL);
}
// Sᵀ block (VGPRs) += K fragment x Q fragment.  As an asm statement with "v" operands: the builtin would let hipcc keep the
// accumulators in AGPRs — a[0:63], on top of the literal Oᵀ accumulators (caught by leetcuda_amd/isa_audit.py rule R1).
// FIRST: the first MFMA of a chain takes the inline constant 0 as its accumulator input and only WRITES s (round 4: until then hipcc
// zeroed the 16 registers with VALU moves per chain and tile, and the statement needed two wait states in front — isa_audit.py rule R6).
template <bool BF16, bool FIRST = false>
LC_DEVINL void bd2_qk(f32x16_t& s, half8_t k, half8_t q) {
#define LC_BD2_QK(OP)                                                                                         \
  if constexpr (FIRST) asm volatile(OP " %0, %1, %2, 0" : "=&v"(s) : "v"(k), "v"(q) : LC_AGPR_ALL);             \
  else asm volatile(OP " %0, %1, %2, %0" : "+v"(s) : "v"(k), "v"(q) : LC_AGPR_ALL)
  if constexpr (BF16) { LC_BD2_QK("v_mfma_f32_32x32x16_bf16"); }
  else { LC_BD2_QK("v_mfma_f32_32x32x16_f16"); }
#undef LC_BD2_QK
}
// four of them (d tiles 4dq .. 4dq+3, one P fragment) in ONE statement: hipcc pads a wait state at every asm boundary
template <int R0, bool BF16>
LC_DEVINL void bd2_pv4(half8_t v0, half8_t v1, half8_t v2, half8_t v3, half8_t p) {
#define LC_BD2_PV4(OP)                                                                                               \
  asm volatile(OP " a[%5:%6], %0, %4, a[%5:%6]\n\t" OP " a[%7:%8], %1, %4, a[%7:%8]\n\t" OP                         \
               " a[%9:%10], %2, %4, a[%9:%10]\n\t" OP " a[%11:%12], %3, %4, a[%11:%12]"                              \
               :: "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(p), "n"(R0), "n"(R0 + 15), "n"(R0 + 16), "n"(R0 + 31),      \
                  "n"(R0 + 32), "n"(R0 + 47), "n"(R0 + 48), "n"(R0 + 63) : LC_AGPR_ALL)
  if constexpr (BF16) LC_BD2_PV4("v_mfma_f32_32x32x16_bf16");
  else LC_BD2_PV4("v_mfma_f32_32x32x16_f16");
#undef LC_BD2_PV4
}
// P·V step with the Vᵀ fragments in FOUR FIXED register quads v[240:255] (physical-register constraints): the statement
// waits for fragment j (counted lgkmcnt: LDS returns in order), issues its MFMA and — RD — at once the two transpose reads of
// the NEXT step's fragment j into the same quad (the MFMA has read its operands long before the LDS data lands), so every
// read has a full step (>= 128 matrix-core cycles) of flight and no second register set is needed.
// Entry: 8 reads outstanding, in fragment order (bd2_rd0 or the previous step).  !RD: last step, waits 6 / 4 / 2 / 0.
// The leading s_nop 1: hipcc packs the P fragment with v_cvt_pk right in front of the statement that consumes it, and a
// VALU write needs two wait states before an MFMA reads the register (rule R6 of the ISA audit).
template <int R0, bool BF16, bool RD, int OFF, int HOFF>
LC_DEVINL void bd2_pv4_fix(half8_t& f0, half8_t& f1, half8_t& f2, half8_t& f3, half8_t p, const uint32_t (&vx)[4]) {
#define LC_BD2_STEP(OP, W0, W1, W2, W3, R0_, R1_, R2_, R3_)                                                            \
  asm volatile("s_nop 1\n\ts_waitcnt lgkmcnt(" #W0 ")\n\t" OP " a[%9:%10], v[240:243], %4, a[%9:%10]\n\t" R0_                          \
               "s_waitcnt lgkmcnt(" #W1 ")\n\t" OP " a[%11:%12], v[244:247], %4, a[%11:%12]\n\t" R1_                        \
               "s_waitcnt lgkmcnt(" #W2 ")\n\t" OP " a[%13:%14], v[248:251], %4, a[%13:%14]\n\t" R2_                        \
               "s_waitcnt lgkmcnt(" #W3 ")\n\t" OP " a[%15:%16], v[252:255], %4, a[%15:%16]\n\t" R3_                        \
               : "+{v[240:243]}"(f0), "+{v[244:247]}"(f1), "+{v[248:251]}"(f2), "+{v[252:255]}"(f3)                        \
               : "v"(p), "v"(vx[0]), "v"(vx[1]), "v"(vx[2]), "v"(vx[3]), "n"(R0), "n"(R0 + 15), "n"(R0 + 16), "n"(R0 + 31),  \
                 "n"(R0 + 32), "n"(R0 + 47), "n"(R0 + 48), "n"(R0 + 63), "n"(OFF), "n"(OFF + HOFF)                         \
               : LC_AGPR_ALL)
#define LC_BD2_RDS(OP)                                                                                                 \
  LC_BD2_STEP(OP, 6, 6, 6, 6, "ds_read_b64_tr_b16 v[240:241], %5 offset:%17\n\tds_read_b64_tr_b16 v[242:243], %5 offset:%18\n\t", \
              "ds_read_b64_tr_b16 v[244:245], %6 offset:%17\n\tds_read_b64_tr_b16 v[246:247], %6 offset:%18\n\t",             \
              "ds_read_b64_tr_b16 v[248:249], %7 offset:%17\n\tds_read_b64_tr_b16 v[250:251], %7 offset:%18\n\t",             \
              "ds_read_b64_tr_b16 v[252:253], %8 offset:%17\n\tds_read_b64_tr_b16 v[254:255], %8 offset:%18")
  if constexpr (RD) {
    if constexpr (BF16) LC_BD2_RDS("v_mfma_f32_32x32x16_bf16");
    else LC_BD2_RDS("v_mfma_f32_32x32x16_f16");
  } else {
    if constexpr (BF16) LC_BD2_STEP("v_mfma_f32_32x32x16_bf16", 6, 4, 2, 0, "", "", "", "");
    else LC_BD2_STEP("v_mfma_f32_32x32x16_f16", 6, 4, 2, 0, "", "", "", "");
  }
#undef LC_BD2_RDS
#undef LC_BD2_STEP
}
// step 0's fragments: eight transpose reads, in fragment order, into the fixed quads
template <int HOFF>
LC_DEVINL void bd2_rd0(half8_t& f0, half8_t& f1, half8_t& f2, half8_t& f3, const uint32_t (&vx)[4]) {
  asm volatile("ds_read_b64_tr_b16 v[240:241], %4\n\tds_read_b64_tr_b16 v[242:243], %4 offset:%8\n\t"
               "ds_read_b64_tr_b16 v[244:245], %5\n\tds_read_b64_tr_b16 v[246:247], %5 offset:%8\n\t"
               "ds_read_b64_tr_b16 v[248:249], %6\n\tds_read_b64_tr_b16 v[250:251], %6 offset:%8\n\t"
               "ds_read_b64_tr_b16 v[252:253], %7\n\tds_read_b64_tr_b16 v[254:255], %7 offset:%8"
               : "={v[240:243]}"(f0), "={v[244:247]}"(f1), "={v[248:251]}"(f2), "={v[252:255]}"(f3)
               : "v"(vx[0]), "v"(vx[1]), "v"(vx[2]), "v"(vx[3]), "n"(HOFF));
}
// ---- V handed over TRANSPOSED ([D][N]: the reference's *_swizzle_qkv entries): the tile image is [D rows][64 kv] and — with the K rows
// of a 16-block fed to the MFMA in the order pi(m) = m with bits 2 and 3 swapped, so that a lane's eight k slots of P fragment g are
// the CONTIGUOUS kv 16 g + 8 hi .. + 7 — a Vᵀ fragment is ONE ds_read_b128 (granule 2 g + hi of d-row 32 dt + l32) where the [N][D]
// image needs two transpose reads.  Same fixed quads, same step structure; 4 reads outstanding at entry, waits 3 / 3 / 3 / 3
// (last step 3 / 2 / 1 / 0).  A: LDS address of the NEXT step's k-step g (vxg[g1]); O0: its first d tile's byte offset (+ 4 KiB per tile).
template <int R0, bool BF16, bool RD, int O0>
LC_DEVINL void bd2_pv4_fix_vt(half8_t& f0, half8_t& f1, half8_t& f2, half8_t& f3, half8_t p, uint32_t a_next) {
#define LC_BD2_STEPV(OP, W0, W1, W2, W3, R0_, R1_, R2_, R3_)                                                            \
  asm volatile("s_nop 1\n\ts_waitcnt lgkmcnt(" #W0 ")\n\t" OP " a[%6:%7], v[240:243], %4, a[%6:%7]\n\t" R0_                          \
               "s_waitcnt lgkmcnt(" #W1 ")\n\t" OP " a[%8:%9], v[244:247], %4, a[%8:%9]\n\t" R1_                        \
               "s_waitcnt lgkmcnt(" #W2 ")\n\t" OP " a[%10:%11], v[248:251], %4, a[%10:%11]\n\t" R2_                        \
               "s_waitcnt lgkmcnt(" #W3 ")\n\t" OP " a[%12:%13], v[252:255], %4, a[%12:%13]\n\t" R3_                        \
               : "+{v[240:243]}"(f0), "+{v[244:247]}"(f1), "+{v[248:251]}"(f2), "+{v[252:255]}"(f3)                        \
               : "v"(p), "v"(a_next), "n"(R0), "n"(R0 + 15), "n"(R0 + 16), "n"(R0 + 31), "n"(R0 + 32), "n"(R0 + 47), "n"(R0 + 48),  \
                 "n"(R0 + 63), "n"(O0), "n"(O0 + 4096), "n"(O0 + 8192), "n"(O0 + 12288)                                     \
               : LC_AGPR_ALL)
  if constexpr (RD) {
    if constexpr (BF16)
      LC_BD2_STEPV("v_mfma_f32_32x32x16_bf16", 3, 3, 3, 3, "ds_read_b128 v[240:243], %5 offset:%14\n\t", "ds_read_b128 v[244:247], %5 offset:%15\n\t",
                   "ds_read_b128 v[248:251], %5 offset:%16\n\t", "ds_read_b128 v[252:255], %5 offset:%17");
    else
      LC_BD2_STEPV("v_mfma_f32_32x32x16_f16", 3, 3, 3, 3, "ds_read_b128 v[240:243], %5 offset:%14\n\t", "ds_read_b128 v[244:247], %5 offset:%15\n\t",
                   "ds_read_b128 v[248:251], %5 offset:%16\n\t", "ds_read_b128 v[252:255], %5 offset:%17");
  } else {
    if constexpr (BF16) LC_BD2_STEPV("v_mfma_f32_32x32x16_bf16", 3, 2, 1, 0, "", "", "", "");
    else LC_BD2_STEPV("v_mfma_f32_32x32x16_f16", 3, 2, 1, 0, "", "", "", "");
  }
#undef LC_BD2_STEPV
}
// step 0's fragments (k-step 0, d tiles 0 .. 3): four ds_read_b128, in fragment order, into the fixed quads
LC_DEVINL void bd2_rd0_vt(half8_t& f0, half8_t& f1, half8_t& f2, half8_t& f3, uint32_t a0) {
  asm volatile("ds_read_b128 v[240:243], %4\n\tds_read_b128 v[244:247], %4 offset:4096\n\t"
               "ds_read_b128 v[248:251], %4 offset:8192\n\tds_read_b128 v[252:255], %4 offset:12288"
               : "={v[240:243]}"(f0), "={v[244:247]}"(f1), "={v[248:251]}"(f2), "={v[252:255]}"(f3)
               : "v"(a0));
}
template <int OFF>
LC_DEVINL half4_t bd2_tr(uint32_t addr) {   // asm transpose read (hipcc would guard the builtin with vmcnt(0) after LDS-DMA)
  half4_t r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}

template <int D, bool BF16, bool VT = false>
__global__ __launch_bounds__(256) void attn_fwd_bigd2_kernel(
    const half_t* __restrict__ Q, const half_t* __restrict__ K, const half_t* __restrict__ V,
    half_t* __restrict__ O, int N, int nqb, float sl2) {
  static_assert(D == 256 || D == 512, "bigd2: D = 256 or 512");
  constexpr int ROWB = D * 2;              // bytes per K / V row
  constexpr int TILE = KVB * ROWB;         // one K or V tile
  constexpr int NKS = D / 16;              // k-steps of Q·Kᵀ
  constexpr int NDT = D / 32;              // 32-column Oᵀ blocks
  constexpr int CPR = ROWB / 16;           // 16-B chunks per row (64 at D = 512)
  constexpr int PPR = ROWB / 1024;         // DMA pieces per row: 1 (D = 512); D = 256: one piece = 2 rows
  constexpr int NPIECE = TILE / 1024 / 4;  // DMA pieces per wave and tile (16 at D = 512, 8 at D = 256)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = wave_id();
  const int hi = lane >> 5, l32 = lane & 31;

  // (wave-uniform values pinned to SGPRs: as per-lane 64-bit values the head offset and the O pointer would be the first
  // things hipcc spills across the loop — the kernel runs at the 256-VGPR limit — and the ISA audit allows no scratch)
  const int id = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x));
  const int bhi = __builtin_amdgcn_readfirstlane(id / nqb);
  const size_t bh = (size_t)bhi;
  const int q0 = __builtin_amdgcn_readfirstlane((id - bhi * nqb) * 128 + wave * 32);
  const half_t* Qb = Q + bh * (size_t)N * D;
  const char* Kb = (const char*)(K + bh * (size_t)N * D);
  const char* Vb = (const char*)(V + bh * (size_t)N * D);
  half_t* Ob = O + bh * (size_t)N * D;
  const int T = N / KVB;
  const uint32_t smem32 = lds_addr32(smem);
  char* const ksm = smem;
  char* const vsm = smem + TILE;

  // ---- LDS-DMA.  D = 512: piece = one 1-KiB row; this wave stages rows wave + 4i.  D = 256: piece = two 512-B rows
  // 2p, 2p+1 with p = wave + 4i (lane>>5 selects the row).  Lane chunk slot cs holds source chunk cs ^ key(row).
  const buf_rsrc_t rk = make_rsrc(Kb), rv = make_rsrc(Vb);
  unsigned k_off[4], v_off;
  {
    const int cs = lane & (CPR - 1), rsub = PPR ? 0 : (lane >> 5);
    // row = (wave + 4i) [* 2 + rsub at D = 256]; row & 15 takes 4 values over i -> 4 lane-offset registers
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = PPR ? (wave + 4 * j) : (2 * (wave + 4 * j) + rsub);
      // VT: the K fragment of MFMA row m is read from K row pi(m) (bits 2, 3 swapped, below) with the key of the READING lane, m & 15:
      // the row that lane m reads is pi(m), so row r is filled under key pi(r) & 15 (pi is an involution)
      const int kkey = VT ? (((row & 3) | ((row & 4) << 1) | ((row & 8) >> 1)) & 15) : (row & 15);
      k_off[j] = (unsigned)(rsub * ROWB + ((cs ^ kkey) * 16));
    }
    const int rowv = PPR ? wave : (2 * wave + rsub);      // (row & 3) does not depend on i
    v_off = (unsigned)(rsub * ROWB + ((cs ^ ((rowv & 3) << 2)) * 16));
    if constexpr (VT) {
      // V as [D][N]: piece p = d-rows 8 p .. 8 p + 7 of 128 B (64 kv), 2 N bytes apart; lane -> d-row rr = lane >> 3, LDS granule slot
      // lane & 7 <- source granule (lane & 7) ^ key(row), key = (row >> 1) & 7 = 4 (p & 1) + (rr >> 1), p & 1 = wave & 1
      const int rr = lane >> 3, c8 = lane & 7;
      v_off = (unsigned)((size_t)rr * N * 2 + ((c8 ^ (4 * (wave & 1) + (rr >> 1))) * 16));
    }
  }
  const unsigned v_piece_stride = VT ? (unsigned)(16u * (unsigned)N) : 1024u;   // source bytes between consecutive pieces of a V tile
  constexpr unsigned V_TILE_STRIDE = VT ? 128u : (unsigned)TILE;                 // ... between consecutive V tiles
  auto issue_k = [&](int i, int t) {   // piece i of tile t (clamped) -> K region
    const int te = t < T ? t : T - 1;
    const int p = wave + 4 * i;
    blds16(rk, k_off[i & 3], (unsigned)te * TILE + (unsigned)p * 1024u, ksm + p * 1024);
  };
  auto issue_v = [&](int i, int t) {
    const int te = t < T ? t : T - 1;
    const int p = wave + 4 * i;
    blds16(rv, v_off, (unsigned)te * V_TILE_STRIDE + (unsigned)p * v_piece_stride, vsm + p * 1024);
  };
#pragma unroll
  for (int i = 0; i < NPIECE; ++i) issue_k(i, 0);

  // ---- Q fragments -> registers (once): lane holds Q[q0 + l32][16 ks + 8 hi .. +8]
  constexpr int PARK = bigd2_park_ks<D>(), NRES = NKS - PARK;   // k-steps whose Q fragment is parked in LDS / resident
  half8_t qf[NRES];
#pragma unroll
  for (int ks = 0; ks < NRES; ++ks) qf[ks] = *(const half8_t*)(Qb + (size_t)(q0 + l32) * D + 16 * ks + 8 * hi);
  char* const qpark = smem + 2 * TILE + wave * (PARK * 1024) + lane * 16;   // + 1024 per parked k-step
#pragma unroll
  for (int i = 0; i < PARK; ++i)
    *(half8_t*)(qpark + i * 1024) = *(const half8_t*)(Qb + (size_t)(q0 + l32) * D + 16 * (NRES + i) + 8 * hi);
  static_for<D / 2>([&](auto r) { am_acc_zero<decltype(r)::value>(); });

  // ---- fragment read addresses
  const char* kx[8];   // K: row l32 (+32 tt), chunk (2ks + hi): low 4 bits XOR (row & 15); + (ks >> 3) * 256 as immediate
#pragma unroll
  for (int k8 = 0; k8 < 8; ++k8)
    kx[k8] = ksm + (VT ? ((l32 & 19) | ((l32 & 4) << 1) | ((l32 & 8) >> 1)) : l32) * ROWB + (((2 * k8 + hi) ^ (l32 & 15)) * 16);
  const int vi = lane & 15, vgi = (lane >> 4) & 1;
  uint32_t vx[4];   // Vᵀ: kv row 4hi + (vi>>2) (+16g, +8), 64-B unit dt: low 2 bits XOR (row & 3); + (dt >> 2) * 256 immediate
#pragma unroll
  for (int b = 0; b < 4; ++b)
    vx[b] = VT ? smem32 + (uint32_t)(TILE + l32 * 128 + ((((2 * b) | hi) ^ ((l32 >> 1) & 7)) * 16))     // VT: vx[g] = k-step g's granule 2 g + hi of d-row l32 (+ 4 KiB per d tile)
               : smem32 + (uint32_t)(TILE + (4 * hi + (vi >> 2)) * ROWB + 32 * vgi + 8 * (vi & 3) + ((b ^ (vi >> 2)) << 6));

  float m_run = -INFINITY, l_run = 0.f;
  half8_t pfa[4], pfb[4];   // P fragments (k-step g = 16 kv rows) of the even / odd tiles

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  raw_barrier();   // K(0) landed

  // ---- P·V step st of the tile whose P fragments are `pf`: retire this step's transpose reads, issue the next step's,
  // then the four MFMAs (one statement).  step = (g, dq): k-step g = 16 kv rows, dq = quad of 32-column d tiles (dt = 4dq + j);
  // address: vx[dt & 3] + (dt >> 2) * 256 = vx[j] + dq * 256, + g * 16 rows (+ 8 rows for the second half)
  // The Vᵀ fragments of a step live in ONE register set, the fixed quads v[240:255] (bd2_pv4_fix): a second set would not
  // fit beside Sᵀ, Q and two P sets.
  half8_t vf0, vf1, vf2, vf3;
  constexpr int NQ = NDT / 4, NST = 4 * NQ;
  // DMA piece i of a phase is issued in k-step i * SPAN_A / NPIECE (Q·Kᵀ phase) / step i * SPAN_B / NPIECE (P·V phase)
  // over SP8 eighths of the phase: measured flat from 4/8 to 6/8 at D = 512 and from 3/8 to 4/8 at D = 256, 7/8 resp. 6/8 cost 1 ... 2 %
  // (profiles/r3aa_attn_bigd2_span_sweep.log)
  constexpr int SP8 = D == 512 ? 6 : 4;
  constexpr int SPAN_A = SP8 * NKS / 8, SPAN_B = SP8 * NST / 8;
  auto rd0 = [&]() {
    if constexpr (VT) bd2_rd0_vt(vf0, vf1, vf2, vf3, vx[0]);
    else bd2_rd0<8 * ROWB>(vf0, vf1, vf2, vf3, vx);
  };
  auto pv_step = [&](auto stc, half8_t (&pf)[4]) {
    constexpr int st = decltype(stc)::value, g = st / NQ, dq = st % NQ;
    constexpr int g1 = (st + 1) / NQ, dq1 = (st + 1) % NQ;
    if constexpr (VT) bd2_pv4_fix_vt<64 * dq, BF16, (st + 1 < NST), dq1 * 4 * 4096>(vf0, vf1, vf2, vf3, pf[g], vx[g1 & 3]);
    else bd2_pv4_fix<64 * dq, BF16, (st + 1 < NST), dq1 * 256 + g1 * 16 * ROWB, 8 * ROWB>(vf0, vf1, vf2, vf3, pf[g], vx);
  };

  // ---- one tile period.  Phase A: Sᵀ(t) = K(t)·Qᵀ with the DMA of V(t−1) in its shadow; barrier; phase B: P·V(t−1) with
  // softmax(t) as compiler-scheduled filler between its MFMA statements (2 score elements per step) and the DMA of K(t+1);
  // barrier.  The softmax is off the MFMA critical path that way (it was a serial ~20 % of the tile period).
  // pn = P(t) (written), po = P(t−1) (read).  HAS_PV = false: tile 0.
  auto tile = [&](auto pvc, int t, half8_t (&pn)[4], half8_t (&po)[4]) {
    constexpr bool HAS_PV = decltype(pvc)::value;
    f32x16_t s[2];   // [tt]: two independent accumulation chains (an MFMA depends on the one before the previous); written by k-step 0
    // K fragments: plain LDS loads (hipcc counts their lgkmcnt), software-pipelined by hand TWO k-steps ahead through a
    // ring of three register pairs — behind opaque asm MFMAs hipcc would otherwise load and wait in the same k-step
    {
      half8_t kfr[3][2], qfr[3];
      auto ldk = [&](auto kc, auto rc) {
        constexpr int ks = decltype(kc)::value, r = decltype(rc)::value;
        kfr[r][0] = *(const half8_t*)(kx[ks & 7] + (ks >> 3) * 256);
        kfr[r][1] = *(const half8_t*)(kx[ks & 7] + (ks >> 3) * 256 + 32 * ROWB);
        if constexpr (ks >= NRES) qfr[r] = *(const half8_t*)(qpark + (ks - NRES) * 1024);
      };
      ldk(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
      ldk(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
      static_for<NKS>([&](auto kc) {
        constexpr int ks = decltype(kc)::value;
        if constexpr (ks + 2 < NKS) ldk(std::integral_constant<int, ks + 2>{}, std::integral_constant<int, (ks + 2) % 3>{});
        // V(t−1): two pieces per three k-steps, the last one three quarters into the phase.  The tiles are single-buffered
        // (the phase ends with vmcnt(0) + barrier: a piece issued late exposes its L2 / HBM latency there), but the texture-
        // address unit takes 16 cycles per piece and serves four waves: two pieces per k-step and wave (round 2) asked for
        // twice what it can take, and the waves stood at the full FIFO for 7 % of their cycles with the MFMAs queued behind
        // the DMA (SQ_VMEM_TA_CMD_FIFO_FULL, profiles/r3u_pmc_attn.txt).  Now 2/3 of its rate.
        // (D = 256: the phase is half as long — 1024 matrix-core cycles — and the L2 read latency is not: first half of the phase)
        if constexpr (HAS_PV)
          static_for<NPIECE>([&](auto ic) {
            if constexpr (decltype(ic)::value * SPAN_A / NPIECE == ks) issue_v(decltype(ic)::value, t - 1);
          });
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ks < NRES) {
          bd2_qk<BF16, ks == 0>(s[0], kfr[ks % 3][0], qf[ks]);
          bd2_qk<BF16, ks == 0>(s[1], kfr[ks % 3][1], qf[ks]);
        } else {
          bd2_qk<BF16>(s[0], kfr[ks % 3][0], qfr[ks % 3]);
          bd2_qk<BF16>(s[1], kfr[ks % 3][1], qfr[ks % 3]);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    am_drain(s[0], s[1]);   // asm MFMAs: hipcc does not know their latency; VALU reads S next
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // own V(t−1) pieces landed, own K reads retired
    raw_barrier();                                                // K(t) is dead, V(t−1) complete

    // =========================== phase B
    float ps0 = 0.f, ps1 = 0.f;
    const float nm = -m_run;
    if constexpr (HAS_PV) rd0();
    static_for<NST>([&](auto stc) {
      constexpr int st = decltype(stc)::value;
      // K(t+1): spread the same way over the P·V steps (D = 512: 2, 1, 1 pieces per three steps)
      static_for<NPIECE>([&](auto ic) {
        if constexpr (decltype(ic)::value * SPAN_B / NPIECE == st) issue_k(decltype(ic)::value, t + 1);
      });
      if constexpr (HAS_PV) pv_step(stc, po);
      __builtin_amdgcn_sched_barrier(0);
      // softmax(t) of score elements 32 st / NST .. : row sums from the unrounded P (tiling_qkv.cu keeps the same order)
      static_for<32 / NST>([&](auto jc) {
        constexpr int e = st * (32 / NST) + decltype(jc)::value, tt = e >> 4, r = e & 15;
        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[tt][r], sl2, nm));
        if constexpr ((r & 1) != 0) ps1 += p; else ps0 += p;
        pn[2 * tt + (r >> 3)][r & 7] = cvt16<BF16>(p);
      });
      __builtin_amdgcn_sched_barrier(0);
    });
    float psum = ps0 + ps1;
    if (!__all(psum_below(psum, 16384.0f)) || !HAS_PV) {        // overflow guard / first tile: establish the true max
      float mx = s[0][0];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[tt][r]);
      mx = am_xhalf_max(mx * sl2);                   // (sl2 > 0)
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // exp2(-inf) = 0 on the first tile
      m_run = m_new;
      l_run *= alpha;
      am_drain();    // the P·V MFMAs of this phase have written Oᵀ
      static_for<D / 2>([&](auto rc) { am_acc_scale<decltype(rc)::value>(alpha); });
      psum = 0.f;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[tt][r], sl2, -m_run));
          psum += p;
          pn[2 * tt + (r >> 3)][r & 7] = cvt16<BF16>(p);
        }
    }
    l_run += psum;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // own K(t+1) pieces landed, own V reads retired
    raw_barrier();                                                // V(t−1) is dead, K(t+1) complete
  };
  using HAS = std::integral_constant<bool, true>;
  using HASNOT = std::integral_constant<bool, false>;
  tile(HASNOT{}, 0, pfa, pfb);
  tile(HAS{}, 1, pfb, pfa);
  for (int t = 2; t < T; t += 2) {      // T = N / 64 is even (N % 128 == 0)
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

  // ---- epilogue: O = Oᵀ / l through LDS (whole rows, 16-B stores).  Lane holds O[q = l32][d = 32dt + 8rq + 4hi + (0..3)]
  // in a[16dt + 4rq ..]; every wave owns a private 32 x (ROWB + 16) B staging area (the KV tiles are dead).
  constexpr int ESTR = ROWB + 16;
  am_drain();
  const float inv = 1.0f / am_xhalf_sum(l_run);
  char* stg = smem + wave * (32 * ESTR);
  // the lane id again, from mbcnt: keeping `lane` / `l32` / `hi` alive across the loop costs the registers hipcc would spill
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
  constexpr int LPR = ROWB / 16;             // lanes per row
  constexpr int RPI = 64 / LPR;              // rows per wave-instruction (1 at D = 512, 2 at D = 256)
#pragma unroll
  for (int it = 0; it < 32 / RPI; ++it) {
    const int row = it * RPI + lane_e / LPR, c = lane_e % LPR;
    const u32x4_t v = *(const u32x4_t*)(stg + row * ESTR + c * 16);
    *(u32x4_t*)(ow + (size_t)row * D + c * 8) = v;
  }
}

}  // namespace lc
