// gemm_fp8_w4k.hip — fp8 (OCP e4m3fn) GEMM for gfx950 on v_mfma_scale_f32_16x16x128_f8f6f4: C[M,N] (fp16) = alpha · A8[M,K] · B8ᵀ, B8 stored
// [N,K]; with MX = true every 32 consecutive k of every row of A and B carry their own E8M0 block scale (OCP MX: value = e4m3 · 2^(s − 127)).
//
// BASELINE config 5 ("fp8 MFMA HGEMM M=N=K=16384"); the reference has no fp8 GEMM — an EXTENSION of the path, parity is
// defined against the fp64 oracle on the decoded values.  Structure = hgemm_w4y_kernel (hgemm_w4y.hip) byte for byte outside the K loop:
// 256 x 256 C tile per workgroup, four wave64 with 128 x 128 wave tiles, accumulators a[0:255] as 8 x 8 blocks of 16 x 16, LDS ring of
// A 2 + B 3 K tiles (a 128-byte row now holds 128 k values), LDS-DMA pieces of 8 rows with the XOR swizzle applied on the source side,
// by-XCD K-loop stagger, persistent workgroups with the next C tile's first two K tiles prefetched under the epilogue.  The K loop is ONE
// generated statement (gemm_fp8_w4k_loop{,_mx}.inc, tools/gen_gemm_fp8_w4k.py: 64 MFMAs of K = 128 per K tile, Gray-code walk over the
// wave tile's quadrants so that 128 fragment registers suffice).  gemm_fp8_w4_kernel (gemm_fp8.hip, v_mfma_scale_f32_32x32x64, the
// compiler-scheduled kernel of rounds 2-3) stays as the cross-check: both accumulate exact e4m3 products in fp32, in a different order.
//
// MX operand layout (tools/cpp/mx_probe.cpp, profiles/r4m_mx_probe.log): lane l of the instruction holds row l % 16 with k = 16 (l / 16) + 0..15
// in registers 0-3 and k = 64 + 16 (l / 16) + 0..15 in registers 4-7, and SUPPLIES the scale of row l % 16, k block l / 16 (k = 32 (l / 16) ..
// + 31) in the byte of its scale register that op_sel / op_sel_hi select.  The kernel therefore wants the scales packed (mx_pack_scales):
//   P[rows / 128][K / 128][64 lanes][2] dwords;   byte f of dword (R, t, l, h) = S[128 R + 64 h + 16 f + l % 16][4 t + l / 16]
// so that one buffer_load_dwordx2 per operand, lane and K tile fetches the scale bytes of all eight fragments of that operand.
#pragma once
#include "hgemm_w4y.hip"

namespace lc {

constexpr int BK8K = 128;   // k elements (= bytes) per K tile

// natural [rows][K / 32] E8M0 scales -> the packed layout above; one thread per output dword
__global__ __launch_bounds__(256) void mx_pack_scales_kernel(const uint8_t* __restrict__ S, uint32_t* __restrict__ P, int rows, int K) {
  const int KT = K / BK8K, KB = K / 32;
  const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)(rows / 128) * KT * 128;
  if (id >= total) return;
  const int h = (int)(id & 1), l = (int)((id >> 1) & 63);
  const size_t rt = id >> 7;
  const int t = (int)(rt % KT);
  const size_t R = rt / KT;
  uint32_t w = 0;
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const size_t row = 128 * R + 64 * h + 16 * f + (l & 15);
    w |= (uint32_t)S[row * KB + 4 * t + (l >> 4)] << (8 * f);
  }
  P[id] = w;
}

template <bool MX>
__global__ __launch_bounds__(256) void gemm_fp8_w4k_kernel(const uint8_t* __restrict__ A, const uint8_t* __restrict__ B,
                                                           half_t* __restrict__ C, int M, int N, int K, float alpha,
                                                           int tiles_m, int tiles_n, int panel_w, int stagger, int ntiles,
                                                           const uint32_t* __restrict__ PA, const uint32_t* __restrict__ PB) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = wave_id();
  const int wr = wave >> 1, wc = wave & 1;

  // persistent walk over the virtual block ids as hgemm_w4y_kernel (ntiles > 0: one workgroup per CU)
  int vb = __builtin_amdgcn_readfirstlane((int)blockIdx.x);
  const int vstride = __builtin_amdgcn_readfirstlane((int)gridDim.x);
  const int nvb = ntiles > 0 ? ntiles : vstride;
  const int KT = K / BK8K;

  // DMA: piece g of this wave = rows 32 g + 8 wave .. + 8 of the A / B tile; lane -> row lane >> 3, 16-byte chunk lane & 7 stored at
  // chunk ^ ((row >> 1) & 7), (row >> 1) & 7 = (lane >> 4) & 3 | (wave & 1) << 2 for every piece
  const uint32_t w4k_ao = (uint32_t)(lane >> 3) * (uint32_t)K + (uint32_t)(((lane & 7) ^ (((lane >> 4) & 3) | ((wave & 1) << 2))) * 16);
  const uint32_t blk_bytes = 32u * (uint32_t)K;

  struct TileCtx {
    int m0, n0;
    const char* ua;
    const char* ub;
    uint32_t stg;
  };
  auto tile_ctx = [&](int v) -> TileCtx {
    const TileCoord tc = block_tile(v, nvb, tiles_m, tiles_n, panel_w);
    TileCtx c;
    c.m0 = tc.tm * BM;
    c.n0 = tc.tn * BN;
    c.ua = (const char*)(A + (size_t)(c.m0 + wave * 8) * K);
    c.ub = (const char*)(B + (size_t)(c.n0 + wave * 8) * K);
    const int cx = stagger & 15, cm = (stagger >> 4) & 15, cn = (stagger >> 8) & 15, step = (stagger >> 12) & 0xff, mask = (stagger >> 20) & 0x7f;
    const int idx = cx * (v & 7) + cm * __builtin_amdgcn_readfirstlane(tc.tm) + cn * __builtin_amdgcn_readfirstlane(tc.tn);
    c.stg = (uint32_t)__builtin_amdgcn_readfirstlane((int)((unsigned)((idx & mask) * step) % (unsigned)KT));
    return c;
  };
  auto a_slot = [&](int t) -> char* { return smem + (t & 1) * TILE_BYTES; };
  auto b_slot = [&](int bi) -> char* { return smem + (2 + bi) * TILE_BYTES; };
  auto issue_prologue = [&](const TileCtx& c) {
    const buf_rsrc_t ra = make_rsrc(c.ua), rb = make_rsrc(c.ub);
    auto piece = [&](int g, int t, char* slot) {
      int te = (t < KT ? t : KT - 1) + (int)c.stg;
      if (te >= KT) te -= KT;
      blds16(g < 8 ? ra : rb, w4k_ao, (unsigned)(g & 7) * blk_bytes + (unsigned)te * BK8K, slot + (g & 7) * 4096 + wave * 1024);
    };
#pragma unroll
    for (int g = 8; g < 16; ++g) piece(g, 0, b_slot(0));
#pragma unroll
    for (int g = 0; g < 8; ++g) piece(g, 0, a_slot(0));
#pragma unroll
    for (int g = 8; g < 16; ++g) piece(g, 1, b_slot(1));
#pragma unroll
    for (int g = 0; g < 8; ++g) piece(g, 1, a_slot(1));
  };

  // fragment read lane offsets: register half x of a fragment = 16-byte chunk 4 x + (lane >> 4) of row lane & 15 (+ 16 per fragment)
  W4xFrag fr;
  w4x_frag_init(fr, wr, wc, lane);

  TileCtx cur = tile_ctx(vb);
  issue_prologue(cur);
  for (;;) {
    static_for<256>([&](auto r) { w4_acc_zero<decltype(r)::value>(); });
    LC_VMCNT(16);   // K tile 0 landed (the reasoning for 16: hgemm_w4y.hip)
    pp_barrier();
    const int m0 = cur.m0, n0 = cur.n0;
    const uint32_t w4k_stg = cur.stg;
    {
      const uint64_t pa = (uint64_t)cur.ua, pb = (uint64_t)cur.ub;
      const u32x4_t w4k_ra = {(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)pa),
                              (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(pa >> 32)) & 0xffffu, 0x7fffffffu, 0x00020000u};
      const u32x4_t w4k_rb = {(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)pb),
                              (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(pb >> 32)) & 0xffffu, 0x7fffffffu, 0x00020000u};
      const uint32_t w4k_a0 = __builtin_amdgcn_readfirstlane(lds_addr32(smem));
      const uint32_t w4k_wv = (uint32_t)wave * 1024u;
      const uint32_t w4k_blk = __builtin_amdgcn_readfirstlane(blk_bytes);
      uint32_t w4k_t, w4k_acur, w4k_anxt, w4k_b0, w4k_b1, w4k_b2, w4k_soff, w4k_t2off, w4k_tmp, w4k_swp;
      if constexpr (MX) {
        // packed scales of this wave's 128 A rows / 128 B rows: K tile t at byte offset 512 t, lane part 8 lane
        const uint64_t qa = (uint64_t)(PA + (size_t)((m0 >> 7) + wr) * KT * 128), qb = (uint64_t)(PB + (size_t)((n0 >> 7) + wc) * KT * 128);
        const u32x4_t w4k_rsa = {(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)qa),
                                 (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(qa >> 32)) & 0xffffu, 0x7fffffffu, 0x00020000u};
        const u32x4_t w4k_rsb = {(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)qb),
                                 (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(qb >> 32)) & 0xffffu, 0x7fffffffu, 0x00020000u};
        const uint32_t w4k_slo = (uint32_t)lane * 8u;
        uint32_t w4k_s1off;
#include "gemm_fp8_w4k_loop_mx.inc"
      } else {
        uint32_t w4k_one = 0x7f7f7f7fu;
        asm volatile("" : "+v"(w4k_one));
#include "gemm_fp8_w4k_loop.inc"
      }
    }
    // a K = 128 MFMA is 8 passes; drain, then the ring is dead: prefetch the next C tile's first two K tiles under the epilogue
    w4_mfma_drain();
    w4_mfma_drain();
    __syncthreads();
    const int vbn = __builtin_amdgcn_readfirstlane(vb + vstride);
    const bool has_next = ntiles > 0 && vbn < ntiles;
    TileCtx nxt = cur;
    if (has_next) {
      nxt = tile_ctx(vbn);
      issue_prologue(nxt);
    }
    w4y_epilogue_b2<true>(smem + 4 * TILE_BYTES, C, N, m0, n0, wave, wr, wc, lane, alpha);
    if (!has_next) break;
    vb = vbn;
    cur = nxt;
  }
}

}  // namespace lc
