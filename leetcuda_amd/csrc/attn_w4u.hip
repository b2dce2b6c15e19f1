// attn_w4u.hip — FlashAttention-2 forward, D = 64 / 128, N % 256 == 0 (one block per workgroup: also N % 256 == 128): THE merged-phase 4-wave kernel (round 4: one templated body
// replaces attn_w4g.hip (one block per workgroup), attn_w4p.hip (persistent workgroup), attn_w4n.hip (its D = 128 twin) and the retired
// attn_w4m.hip / attn_w8g.hip; lc_tune_set "attn_nw" = 513 / 515 / 517 select WALK = 0 / 1 / 2).
//
// Same semantics / entry points as attn_fwd.hip (reference: kernels/flash-attn/mma/basic/flash_attn_mma_split_q.cu:55-699,
// dispatcher :769-815; flash_attn_mma_share_qkv.cu:46-769; V handed over as [B,H,D,N]:
// kernels/flash-attn/mma/swizzle/flash_attn_mma_share_qkv_swizzle_qkv.cu:961-1010, flash_attn_mma.py:441-442,716).
//
// template <D, VT, WALK>
//   D     64 or 128 (geometry: attn_mp.h W4G<D>)
//   VT    V is [B,H,D,N]: the tile image in LDS is [D rows][64 kv] (128-B rows, 16-B granule j of row d at slot j ^ ((d >> 1) & 7)),
//         filled by LDS-DMA pieces of 8 d-rows (source stride 2 N bytes), and a Vᵀ fragment is TWO PLAIN ds_read_b64 (kv 4 g .. + 3
//         of kv block 0 / of kv block 1 — the k-slot order the lane-local Pᵀ operand defines) where the [N][D] image needs two
//         ds_read_b64_tr_b16: one for one, same slots, same waits -> the V-transposed entries run at their siblings' speed
//         (round 3: the lock-step kernel, − 27 %).  Conflict-free: tests/test_layouts.py.
//   WALK  0  one 256-row query block per workgroup (grid = #blocks; the hardware dispatches)
//         1  persistent workgroup per CU, static walk w, w + G, w + 2 G, … (round 3's attn_w4p: the K / V / Q streams continue
//            across block seams)
//         2  persistent workgroup per CU, DYNAMIC queue: workgroup on XCD x claims the next id of ITS XCD (x + 8 j) with one
//            atomic per block, one block ahead (issued behind the last P·V MFMAs of block b − 1, broadcast through LDS at block
//            b's prologue barrier: no exposed latency); the counters reset themselves when the last workgroup leaves.
//         3  SPLIT-KV (round 5; grids that do not fill the GPU — the reference author's regime "B <= 4, H <= 48, SeqLen <= 8192",
//            README.md:120): `nsplit` workgroups per 256-row query block, workgroup (block, s) walks the KV tiles
//            [s T / nsplit, (s + 1) T / nsplit) exactly as WALK 0 walks a whole head and writes its NORMALISED partial O (fp16, layout
//            [nsplit][B H][N][D] in the workspace `O` points to) plus the base-2 log-sum-exp of its range per query row
//            (`lse`, [nsplit][B H][N] fp32); attn_split_combine_kernel merges them: O = sum_s 2^(L_s − L) O_s, L = log2 sum_s 2^L_s.
//            (A one-launch form — the last of a query block's nsplit workgroups to arrive merges it, flash-decoding's semaphore — was
//            built and measured in round 5: bit-identical and 2 x slower, one workgroup merging 256 rows is a serial tail of dependent
//            loads against a 4.9 us combine kernel: profiles/r5c_attn_split_fused.log, r5f_small_split_kernel_durations.log; removed.)
// The arithmetic of a block is the same instruction for instruction in all three walks and for both V layouts' Q·Kᵀ / softmax
// (VT changes only where Vᵀ fragments come from): WALK 0 / 1 / 2 are bit-identical to each other (GPU test).
//
// What happens BETWEEN two 256-row query blocks of a persistent workgroup: the LDS-DMA of "tile T" and "tile
// T + 1" stages tiles 0 / 1 of the NEXT block into ring slots 0 / 1 — exactly where the next prologue reads them (T % 4 == 0);
// the next block's Q rows are requested right after the last P·V MFMAs and land during the O epilogue; the O staging area sits
// behind ring slots 0 / 1, so the epilogue never touches the slots being filled.  One extra barrier per block.
#pragma once
#include "attn_mp.h"

namespace lc {

// claim counters of the dynamic walk: 1024 rotating slots of 16 words ([0..7] next j per XCD, [8] workgroups that left); zero at
// module load, every launch leaves its slot zeroed again (the last workgroup out resets it).  The host picks the slot (ticket
// mod 1024): launches in flight on different streams use different slots unless 1024 of them are outstanding at once.
constexpr int W4U_QSLOTS = 1024;
static __device__ unsigned int g_w4u_queue[W4U_QSLOTS][16];   // (one array per translation unit that instantiates the kernel)

template <int D>
struct W4U {
  using G = W4G<D>;
  static constexpr int EPI_OFF = 2 * G::SLOT;                                // staging behind ring slots 0, 1
  static constexpr int EPI_BYTES = 4 * 64 * G::EPI_STRIDE;
  static constexpr int MBOX = (EPI_OFF + EPI_BYTES > G::LDS) ? EPI_OFF + EPI_BYTES : G::LDS;   // 16 B: the claimed next block id
  static constexpr int LDS = MBOX + 16;
  static_assert(LDS <= 160 * 1024, "ring + staging must fit a CU's LDS");
};

// W4U_STAMPS (liblc_diag.so only, csrc/diag/attn_w4u_stamps.hip): s_memtime stamps of wave 0 of workgroup 0 at the milestones of a block,
// parked in 256 B of LDS behind the kernel's own allocation and copied to g_w4u_stamps at the end (tools/attn_w4u_stamps.py: where the
// fixed cost of a block goes).  The production library never defines it.
#ifdef W4U_STAMPS
static __device__ unsigned long long g_w4u_stamps[32];
#define W4U_STAMP(k)                                                                                                       \
  do {                                                                                                                     \
    if (blockIdx.x == 0 && wave == 0 && lane == 0)                                                                         \
      *(volatile unsigned long long*)(smem + W4U<D>::LDS + 8 * (k)) = __builtin_readcyclecounter();                         \
  } while (0)
#else
#define W4U_STAMP(k) do { } while (0)
#endif

template <int D, bool VT, int WALK>
__global__ __launch_bounds__(256) void attn_fwd_w4u_kernel(
    const half_t* __restrict__ Q, const half_t* __restrict__ K, const half_t* __restrict__ V,
    half_t* __restrict__ O, int N, int nqb, float sl2, int nblk, int nwg, int qslot, int nsplit, float* __restrict__ lse) {
  static_assert(D == 64 || D == 128, "merged-phase attention kernel: D = 64 or 128 (D = 96 / 32: attn_w4i.hip)");
  static_assert(WALK >= 0 && WALK <= 3, "WALK: 0 one block per workgroup, 1 static persistent walk, 2 dynamic queue, 3 split-KV");
  constexpr bool PERSIST = WALK == 1 || WALK == 2;
  constexpr bool SPLIT = WALK == 3;
  using G = W4G<D>;
  constexpr int NDS = G::NDS, NDB = G::NDB, ROWB = G::ROWB, TILE = G::TILE, SLOT = G::SLOT, NS = G::NS;
  constexpr int NRV = G::NRV, NRK = G::NRK, PPW = G::PPW, KBUF = G::KBUF;
  constexpr int GO = G::O, GK = G::K, GQ = G::Q;
  constexpr int NQ = 4 * NDS;   // Q fragments (16 bytes each) per lane
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = wave_id();
  const int g4 = lane >> 4, l16 = lane & 15;
  const int T = SPLIT ? N / KVB / nsplit : N / KVB;   // KV tiles this workgroup walks (SPLIT: its share of the head's, >= 2)
  const uint32_t smem32 = lds_addr32(smem);
  const size_t head_elems = (size_t)N * D;
  W4U_STAMP(0);
#ifdef W4U_STAMPS
  if (blockIdx.x == 0 && wave == 0 && lane == 0) *(volatile unsigned long long*)(smem + W4U<D>::LDS + 8 * 14) = __builtin_amdgcn_s_memrealtime();
#endif

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
  if constexpr (VT) {
    // V as [D][N]: piece p = d-rows 8 p .. 8 p + 7 (128 B = 64 kv each, 2 N bytes apart in memory); lane -> d-row rr = lane >> 3,
    // LDS granule slot cs = lane & 7 <- source granule cs ^ key(row), key = (row >> 1) & 7 = 4 (p & 1) + (rr >> 1), p & 1 = wave & 1
    const int rr = lane >> 3, cs = lane & 7;
    v_off = (unsigned)((size_t)rr * N * 2 + ((cs ^ (4 * (wave & 1) + (rr >> 1))) * 16));
  }
  const unsigned v_piece_stride = VT ? (unsigned)(8u * (unsigned)N * 2u) : 1024u;   // source bytes between consecutive pieces of a V tile
  constexpr unsigned V_TILE_STRIDE = VT ? 128u : (unsigned)TILE;                     // ... between consecutive V tiles
  // ---- fragment read offsets inside a ring slot (attn_mp.h)
  uint32_t kx[NDS];
#pragma unroll
  for (int ds = 0; ds < NDS; ++ds)
    kx[ds] = (uint32_t)(l16 * ROWB + (((4 * ds + g4) ^ (D == 128 ? l16 : ((l16 >> 1) & 7))) * 16));
  // Vᵀ fragment reads (8 bytes each).  [N][D] image: transpose reads — kv row 4 g4 + (l16 >> 2) (+16 x, +32 per half-tile:
  // immediates), 8 bytes at column 4 (l16 & 3) of pair db; D = 128: vx[u] addresses pair 2 u (pair 2 u + 1 sits at ±32 B: key bit 0
  // = g4 & 1, not an immediate); D = 64: vx[db].  VT ([D][64 kv] image, 128-B rows): d-row 16 db + l16 (16 db rows = an immediate),
  // kv 32 H + 16 x + 4 g4 .. + 3 = half (g4 & 1) of granule 4 H + 2 x + (g4 >> 1) at slot granule ^ ((l16 >> 1) & 7): the XOR with
  // 4 H + 2 x is not an immediate -> vx[2 H + x], four address registers like the D = 128 transposed image.
  constexpr int NVX = (VT || D == 128) ? 4 : NDB;
  uint32_t vx[NVX];
#pragma unroll
  for (int u = 0; u < NVX; ++u) {
    if constexpr (VT)
      vx[u] = (uint32_t)(TILE + l16 * 128 + ((((2 * u) | (g4 >> 1)) ^ ((l16 >> 1) & 7)) * 16) + 8 * (g4 & 1));
    else if constexpr (D == 128)
      vx[u] = (uint32_t)(TILE + (4 * g4 + (l16 >> 2)) * 256 + (((2 * u) ^ (((l16 >> 2) << 1) | (g4 & 1))) * 32) + 8 * (l16 & 3));
    else
      vx[u] = (uint32_t)(TILE + (4 * g4 + (l16 >> 2)) * 128 + ((u ^ (((g4 & 1) << 1) | (l16 >> 3))) * 32) + 8 * (l16 & 3));
  }
  const uint32_t vodd = (uint32_t)((g4 & 1) ? -32 : 32);

  // ---- block walk: virtual block vb -> (head, first query row of this wave)
  int vb = __builtin_amdgcn_readfirstlane((int)blockIdx.x);
  int sp = 0;   // SPLIT: which KV range of the head this workgroup walks (ids of one query block are consecutive: one XCD, one Q in L2)
  auto head_of = [&](int v, int& q0w) -> size_t {
    int id = xcd_remap(v, nblk);
    if constexpr (SPLIT) {
      sp = id % nsplit;
      id /= nsplit;
    }
    const int bh = id / nqb;
    q0w = (id - bh * nqb) * 256 + wave * 64;
    return (size_t)bh;
  };
  int q0;
  size_t bh = head_of(vb, q0);
  // SPLIT: element offsets of this workgroup's first KV row inside the head (K and V as [N][D]: kv0 rows; V as [D][N]: kv0 columns)
  const size_t kv0 = SPLIT ? (size_t)sp * T * KVB : 0;
  const size_t k_base = kv0 * D, v_base = VT ? kv0 : kv0 * D;

  // DMA of one K / V piece of the tile this period stages: descriptor + tile index chosen once per tile period (make_rsrc
  // reads the chosen base through readfirstlane: a descriptor hipcc cannot prove wave-uniform gets a waterfall loop per piece)
  buf_rsrc_t dk = make_rsrc(K + bh * head_elems + k_base), dv = make_rsrc(V + bh * head_elems + v_base);
  unsigned d_so = 0;
  char* d_slot = smem;
  unsigned d_sov = 0;   // (V: te * V_TILE_STRIDE — 128 B per tile when V is [D][N])
  auto issue_piece = [&](int i) {   // i = 0 .. 2 PPW−1: K pieces, then V pieces
    const int p = wave + 4 * (i % PPW);
    if (i < PPW)
      blds16(dk, k_off, d_so + (unsigned)p * 1024u, d_slot + p * 1024);
    else
      blds16(dv, v_off, d_sov + (unsigned)p * v_piece_stride, d_slot + TILE + p * 1024);
  };
  // Q rows of a block as raw fp16 (16 bytes per fragment): requested one block ahead
  half8_t qraw[NQ];
  auto load_q = [&](size_t h, int q0w) {
    const half_t* Qb = Q + h * head_elems;
    static_for<NQ>([&](auto ic) {
      constexpr int i = decltype(ic)::value, qb = i / NDS, ds = i % NDS;
      // (row clamp: N % 256 == 128 — legal in the reference, flash_attn_mma_share_qkv.cu:839 — gives the head's last query block 128 real
      // rows; its waves 2 / 3 walk the KV tiles on a copy of row N − 1 and store nothing)
      qraw[i] = *(const half8_t*)(Qb + (size_t)min(q0w + 16 * qb + l16, N - 1) * D + 32 * ds + 8 * g4);
    });
  };

  // dynamic walk: this workgroup's XCD (workgroups are dealt to the XCDs round-robin: id & 7) and its first claim — the id of the
  // block AFTER the first one — in flight behind the first tiles' DMA
  const int xcd = __builtin_amdgcn_readfirstlane((int)blockIdx.x & 7);
  unsigned int* const queue = &g_w4u_queue[qslot][0];
  unsigned claim = 0;
  auto claim_next = [&]() {
    if constexpr (WALK == 2) {
      if (wave == 0 && lane == 0) claim = atomicAdd(queue + xcd, 1u);
    }
  };
  claim_next();
  // seam synchronisation: this block's tiles 0, 1 landed (own pieces; the previous block's O stores too); every wave has left the
  // previous block's staging area.  The dynamic walk also passes the claimed id through the LDS mailbox here (wave 0 writes in
  // front of the barrier, every wave reads behind it) and therefore synchronises at the TOP of a block, before the id is needed.
  auto seam_sync = [&]() {
    if constexpr (WALK == 2) {
      if (wave == 0 && lane == 0) *(volatile unsigned*)(smem + W4U<D>::MBOX) = claim;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    raw_barrier();
  };

  // first block: tiles 0, 1 and Q from scratch.  (Round 5 tried Q FIRST — vector-memory loads return in order, so Q arrives behind the 64 KiB
  // of tiles 0 / 1 — and measured nothing: entry -> "tiles landed" 8500 vs 8716 cycles on an idle GPU, level with split-KV, longer on a full
  // one, where issuing the 16 DMA pieces behind 16 Q loads takes longer: tools/attn_w4u_stamps.py, profiles/r5n_w4u_stamps.log.)
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    d_so = (unsigned)t * TILE;
    d_sov = (unsigned)t * V_TILE_STRIDE;
    d_slot = smem + t * SLOT;
#pragma unroll
    for (int i = 0; i < 2 * PPW; ++i) issue_piece(i);
  }
  load_q(bh, q0);
  W4U_STAMP(1);     // tiles 0, 1 and Q requested

  for (;;) {
    // ---- the block after this one (or this one again when there is none: every address stays valid, nothing of it is used)
    int vbn_;
    if constexpr (WALK == 2) {
      seam_sync();
      vbn_ = nwg + xcd + 8 * (int)*(volatile unsigned*)(smem + W4U<D>::MBOX);   // ids >= nwg with id & 7 == xcd, in claim order
    } else {
      vbn_ = vb + nwg;   // (nwg = gridDim.x as a kernel argument: provably wave-uniform, the block walk stays in SGPRs)
    }
    const int vbn = __builtin_amdgcn_readfirstlane(vbn_);
    const bool has_next = PERSIST && vbn < nblk;
    int q0n;
    const size_t bhn = head_of(has_next ? vbn : vb, q0n);
    const size_t nbh = SPLIT ? (size_t)(nblk / (nqb * nsplit)) : 0;          // B H (SPLIT: partial s of head bh = slab s nbh + bh)
    half_t* Ob = O + ((size_t)sp * nbh + bh) * head_elems;
    // tile t2 >= T of this block = tile t2 − T of the next one (no next block: the last tile again, into a dead slot)
    auto set_dma_tile = [&](int t2) {
      const bool own = t2 < T;
      const size_t h = own ? bh : bhn;
      dk = make_rsrc(K + h * head_elems + k_base);
      dv = make_rsrc(V + h * head_elems + v_base);
      const int te = own ? t2 : (has_next ? t2 - T : T - 1);
      d_so = (unsigned)__builtin_amdgcn_readfirstlane(te * TILE);   // (provably wave-uniform: no waterfall loop around the pieces)
      d_sov = (unsigned)__builtin_amdgcn_readfirstlane(te * (int)V_TILE_STRIDE);
      d_slot = smem + (t2 & 3) * SLOT;
    };

    // ---- Q~ = fp16(Q * scale*log2e) -> AGPRs
    static_for<NQ>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const half8_t q = qraw[i];
      half8_t qs;
#pragma unroll
      for (int e = 0; e < 8; ++e) qs[e] = (half_t)((float)q[e] * sl2);
      const u32x4_t w = __builtin_bit_cast(u32x4_t, qs);
      am_acc_write<GQ + 4 * i + 0>(w[0]);
      am_acc_write<GQ + 4 * i + 1>(w[1]);
      am_acc_write<GQ + 4 * i + 2>(w[2]);
      am_acc_write<GQ + 4 * i + 3>(w[3]);
    });
    static_for<16 * NDB>([&](auto r) { am_acc_zero<decltype(r)::value>(); });
    W4U_STAMP(2);   // Q arrived, converted, parked in AGPRs; O zeroed

    uint32_t ka[NDS], vc[NVX], vp[NVX];
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

    f32x4_t sA[2][4], sB[2][4];
    f32x4_t negm[4];
    half8_t pA[4], pB[4];
    half4_t vlo[NDB], vhi[NDB];
    float l_run[4] = {0.f, 0.f, 0.f, 0.f};

    auto read_k_all = [&](auto bufc, uint32_t sbase, auto hc) {
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
    auto wait_vset = [&](auto firstc) {
      constexpr int first = decltype(firstc)::value;
      if constexpr (NDB == 8) am_wait_v8(reinterpret_cast<half4_t(&)[4]>(vlo[first]), reinterpret_cast<half4_t(&)[4]>(vhi[first]));
      else w4g_wait_v4(vlo[first], vlo[first + 1], vhi[first], vhi[first + 1]);
    };

    // ---- prologue (seam_sync: at the top of the block for the dynamic walk)
    if constexpr (WALK != 2) seam_sync();
    W4U_STAMP(3);   // tiles 0, 1 landed, barrier passed
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using IB = std::integral_constant<int, NDB / 2>;
    read_k_all(I0{}, smem32, I0{});
    read_k_all(I1{}, smem32, I1{});
    am_lgkm0();
    static_for<8 * NDS>([&](auto ic) {
      constexpr int i = decltype(ic)::value, ds = i >> 3, kvb = (i >> 2) & 1, qb = i & 3;
      if constexpr (ds == 0) an_qk_zero<GK + 4 * (NDS * kvb + ds), GQ + 4 * (NDS * qb + ds)>(sA[kvb][qb]);
      else an_qk<GK + 4 * (NDS * kvb + ds), GQ + 4 * (NDS * qb + ds)>(sA[kvb][qb]);
    });
    am_drain(sA);
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) {
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

    // ---- one merged phase ( F bit 8 = issue this period's DMA pieces, target set by set_dma_tile)
    auto phase = [&](auto hc, auto fc, int t, f32x4_t (&sr)[2][4], f32x4_t (&sw)[2][4], half8_t (&pw)[4], half8_t (&pr)[4]) {
      constexpr int H = decltype(hc)::value, F = decltype(fc)::value;
      constexpr bool HAS_PV = (F & 1) != 0, HAS_QK = (F & 2) != 0, HAS_KRD = (F & 4) != 0, HAS_DMA = (F & 8) != 0;
      constexpr int KQ = GK + KBUF * (1 - H);
      constexpr int KRB = H;
      uint32_t (&vb_a)[NVX] = H == 0 ? vp : vc;
      constexpr int VB_H = H == 0 ? 1 : 0;
      wait_vset(I0{});
      float ps[4][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
      float e0 = 0.f, e1 = 0.f, c0 = 0.f, c1 = 0.f;
      auto pair_sum = [&](auto pc, auto wc, float a) {
        constexpr int qb = (decltype(pc)::value >> 1) & 3, w = decltype(wc)::value;
        ps[qb][w] += a;
        asm volatile("" : "+v"(ps[qb][w]));
      };
      auto pair_pack = [&](auto pc, float a, float b) {
        constexpr int p = decltype(pc)::value, kvb = p >> 3, qb = (p >> 1) & 3, k2 = p & 1;
        half2_t h2 = {(half_t)a, (half_t)b};
        asm volatile("" : "+v"(h2));
        pw[qb][4 * kvb + 2 * k2] = h2[0];
        pw[qb][4 * kvb + 2 * k2 + 1] = h2[1];
      };
      static_for<NS>([&](auto sc) {
        constexpr int s = decltype(sc)::value, i = s >> 1;
        constexpr bool RVB = s < NRV && HAS_PV, RK = s < NRK && HAS_KRD, RVA = s >= NS / 2 && s < NS / 2 + NRV;
        constexpr int RD = ((RVB || RVA) ? 1 : 0) | (RK ? 2 : 0);
        constexpr int c = RVA ? s - NS / 2 : (s % NRV), rdb = (RVA ? 0 : NDB / 2) + (c >> 1), rx = c & 1;
        constexpr int VH = RVA ? H : VB_H;     // half-tile of the Vᵀ rows this read fetches, inside their tile
        // [N][D] image: address register by column block, half-tile / kv block as immediates; [D][64 kv] image (VT): address register
        // by (half-tile, kv block) — the granule XOR —, the column block's 16 d-rows (2 KiB) as the immediate
        constexpr int VOF = VT ? rdb * 2048 : VH * 32 * ROWB + rx * 16 * ROWB;
        half4_t& vout = rx ? vhi[rdb] : vlo[rdb];
        const uint32_t vaddr = VT ? (RVA ? vc : vb_a)[(2 * VH + rx) % NVX] : (RVA ? vaddr_of(vc, rdb) : vaddr_of(vb_a, rdb));
        constexpr int kc = s % NRK;
        constexpr int KR = GK + KBUF * KRB + 4 * kc, KOF = H * 32 * ROWB + (kc / NDS) * 16 * ROWB;
        if constexpr ((s & 1) == 0) {
          constexpr int ds = i >> 3, kvb = (i >> 2) & 1, qb = i & 3;
          constexpr int KIND = HAS_QK ? (ds == 0 ? 0 : 1) : 3;
          if constexpr (KIND != 3 || RD != 0)
            an_slot<KIND, RD, KQ + 4 * (NDS * kvb + ds), GQ + 4 * (NDS * qb + ds), VOF, KR, KOF, VT>(
                sw[kvb][qb], negm[qb], half8_t{}, half8_t{}, vout, vaddr, ka[kc % NDS]);
        } else {
          constexpr int db = i >> 2, qb = i & 3;
          constexpr int KIND = HAS_PV ? 2 : 3;
          if constexpr (KIND != 3 || RD != 0)
            an_slot<KIND, RD, GO + 4 * (4 * db + qb), 0, VOF, KR, KOF, VT>(sw[0][0], negm[0], cat4(vlo[db], vhi[db]), pr[qb], vout,
                                                                           vaddr, ka[kc % NDS]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (s == NS / 2 - 1 && HAS_PV) wait_vset(IB{});
        if constexpr (HAS_DMA && (s & 7) == 7) issue_piece(s >> 3);
        if constexpr (NS == 64) {
          if constexpr ((s & 3) == 0) {
            constexpr int p = s >> 2, kvb = p >> 3, qb = (p >> 1) & 3, k2 = p & 1;
            if constexpr (p >= 1) {
              pair_sum(std::integral_constant<int, p - 1>{}, I0{}, e0);
              c0 = e0;
              c1 = e1;
              __builtin_amdgcn_sched_barrier(0);
            }
            e0 = __builtin_amdgcn_exp2f(sr[kvb][qb][2 * k2]);
            asm volatile("" : "+v"(e0));
          } else if constexpr ((s & 3) == 1) {
            if constexpr (s >= 5) pair_sum(std::integral_constant<int, (s >> 2) - 1>{}, I1{}, c1);
          } else if constexpr ((s & 3) == 2) {
            constexpr int p = s >> 2, kvb = p >> 3, qb = (p >> 1) & 3, k2 = p & 1;
            if constexpr (p >= 1) {
              pair_pack(std::integral_constant<int, p - 1>{}, c0, c1);
              __builtin_amdgcn_sched_barrier(0);
            }
            e1 = __builtin_amdgcn_exp2f(sr[kvb][qb][2 * k2 + 1]);
            asm volatile("" : "+v"(e1));
          }
        } else {
          constexpr int p = s >> 1, kvb = p >> 3, qb = (p >> 1) & 3, k2 = p & 1;
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
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      pair_sum(std::integral_constant<int, 15>{}, I0{}, e0);
      pair_sum(std::integral_constant<int, 15>{}, I1{}, e1);
      pair_pack(std::integral_constant<int, 15>{}, e0, e1);
      // ---------------- overflow guard (attn_mp.h)
      uint32_t worst_bits = 0;
#pragma unroll
      for (int qb = 0; qb < 4; ++qb) worst_bits = max(worst_bits, __builtin_bit_cast(uint32_t, ps[qb][0] + ps[qb][1]));
      const bool ok = worst_bits < __builtin_bit_cast(uint32_t, AM_PSUM_LIMIT);
      if (!__all(ok)) {
        am_drain(sw);
        {
          float worst = 0.f;
          bool fin = true;
#pragma unroll
          for (int qb = 0; qb < 4; ++qb) {
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
        static_for<4>([&](auto qc) {
          constexpr int qb = decltype(qc)::value;
          float mx = sr[0][qb][0];
#pragma unroll
          for (int kvb = 0; kvb < 2; ++kvb)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sr[kvb][qb][r]);
          mx = an_x4_max(mx);
          const float delta = fmaxf(mx, 0.f);
          const float alpha = __builtin_amdgcn_exp2f(-delta);
          l_run[qb] *= alpha;
          ps[qb][0] = 0.f;
          ps[qb][1] = 0.f;
#pragma unroll
          for (int kvb = 0; kvb < 2; ++kvb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if constexpr (HAS_QK) sw[kvb][qb][r] -= delta;
              const float pv = __builtin_amdgcn_exp2f(sr[kvb][qb][r] - delta);
              ps[qb][r & 1] += pv;
              pw[qb][4 * kvb + r] = (half_t)pv;
            }
#pragma unroll
          for (int r = 0; r < 4; ++r) negm[qb][r] -= delta;
          static_for<NDB>([&](auto dc) {
            constexpr int db = decltype(dc)::value;
            static_for<4>([&](auto rc) { am_acc_scale<GO + 4 * (4 * db + qb) + decltype(rc)::value>(alpha); });
          });
        });
        asm volatile("s_nop 3" ::: "memory");
      }
#pragma unroll
      for (int qb = 0; qb < 4; ++qb) l_run[qb] += ps[qb][0] + ps[qb][1];
    };
    using F_FIRST0 = std::integral_constant<int, 2 | 4 | 8>;
    using F_MID = std::integral_constant<int, 1 | 2 | 4 | 8>;
    using F_MID1 = std::integral_constant<int, 1 | 2 | 4>;
    using F_LAST0 = std::integral_constant<int, PERSIST ? (1 | 2 | 8) : (1 | 2)>;   // j = 2T−2: stages "tile T + 1" = the next block's tile 1 (one block per workgroup: nothing left to stage)
    using F_LAST1 = std::integral_constant<int, 1>;

    W4U_STAMP(4);   // first S^T (no P.V to overlap), row max
    set_tile_addrs(0);
    set_dma_tile(2);
    phase(I0{}, F_FIRST0{}, 0, sA, sB, pA, pB);
    phase(I1{}, F_MID1{}, 0, sB, sA, pB, pA);
    W4U_STAMP(5);   // tile 0's two phases
    for (int t = 1; t + 1 < T; ++t) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      raw_barrier();
      set_tile_addrs(t);
      set_dma_tile(t + 2);
      phase(I0{}, F_MID{}, t, sA, sB, pA, pB);
      phase(I1{}, F_MID1{}, t, sB, sA, pB, pA);
    }
    W4U_STAMP(6);   // tiles 1 .. T - 2
    {
      const int t = T - 1;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      raw_barrier();
      set_tile_addrs(t);
      set_dma_tile(t + 2);
      phase(I0{}, F_LAST0{}, t, sA, sB, pA, pB);
      phase(I1{}, F_LAST1{}, t, sB, sA, pB, pA);
      static_for<NRV>([&](auto cc) {
        constexpr int c = decltype(cc)::value, db = NDB / 2 + (c >> 1);
        if constexpr (VT) {
          if constexpr ((c & 1) == 0) vlo[db] = lds_rd64_asm<db * 2048>(vc[2 % NVX]);
          else vhi[db] = lds_rd64_asm<db * 2048>(vc[3 % NVX]);
        } else {
          if constexpr ((c & 1) == 0) vlo[db] = lds_tr16_asm<32 * ROWB>(vaddr_of(vc, db));
          else vhi[db] = lds_tr16_asm<32 * ROWB + 16 * ROWB>(vaddr_of(vc, db));
        }
      });
      wait_vset(I0{});
      wait_vset(IB{});
      static_for<4 * NDB>([&](auto ic) {
        constexpr int i = decltype(ic)::value, db = i >> 2, qb = i & 3;
        an_pv<GO + 4 * (4 * db + qb)>(cat4(vlo[db], vhi[db]), pB[qb]);
      });
    }
    W4U_STAMP(7);   // last tile + the P.V that has no Q.K^T to overlap
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PERSIST) {
      if (has_next) claim_next();      // dynamic walk: the id of the block after next (consumed at the next block's seam)
      load_q(bhn, q0n);                // the next block's Q rows: in flight during the epilogue
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue: O = Oᵀ / l through LDS (whole rows, 16-B stores); staging behind ring slots 0 / 1
    am_drain();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    raw_barrier();                     // every wave is done with ring slots 2, 3
    W4U_STAMP(8);   // MFMAs drained, epilogue barrier
    float inv[4];
#pragma unroll
    for (int qb = 0; qb < 4; ++qb) {
      const float lsum = an_x4_sum(l_run[qb]);
      inv[qb] = 1.0f / lsum;   // (IEEE division on purpose: attn_w4i.hip computes the same bits, and the tests compare the two kernels bit for bit)
      if constexpr (SPLIT) {   // base-2 log-sum-exp of this KV range for query row q0 + 16 qb + l16 (scores carry scale * log2 e already)
        if (g4 == 0) lse[((size_t)sp * nbh + bh) * N + q0 + 16 * qb + l16] = __builtin_log2f(lsum) - negm[qb][0];
      }
    }
    char* stg = smem + W4U<D>::EPI_OFF + wave * (64 * G::EPI_STRIDE);
    static_for<4>([&](auto qc) {
      constexpr int qb = decltype(qc)::value;
      // sixteen accumulators per asm statement (lc_common.h acc_read16; round 5: the one-read-per-statement form was a serial chain of
      // ~ 15 dependent instructions per four values, 3600 - 4300 cycles per block — tools/attn_w4u_stamps.py)
      static_for<NDB / 4>([&](auto dc) {
        constexpr int d0 = 4 * decltype(dc)::value;
        float x[16];
        acc_read16<GO + 4 * (4 * d0 + qb), GO + 4 * (4 * (d0 + 1) + qb), GO + 4 * (4 * (d0 + 2) + qb), GO + 4 * (4 * (d0 + 3) + qb)>(x);
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) {
          half4_t h;
          h[0] = (half_t)(x[4 * dd + 0] * inv[qb]);
          h[1] = (half_t)(x[4 * dd + 1] * inv[qb]);
          h[2] = (half_t)(x[4 * dd + 2] * inv[qb]);
          h[3] = (half_t)(x[4 * dd + 3] * inv[qb]);
          *(half4_t*)(stg + (16 * qb + l16) * G::EPI_STRIDE + (16 * (d0 + dd) + 4 * g4) * 2) = h;
        }
      });
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    W4U_STAMP(9);   // O normalised, converted, staged in LDS
    half_t* ow = Ob + (size_t)q0 * D;
    constexpr int LPR = ROWB / 16, RPI = 64 / LPR;
    if (q0 < N) {   // (wave-uniform; false only for waves 2 / 3 of the last query block when N % 256 == 128)
#pragma unroll
      for (int it = 0; it < 64 / RPI; ++it) {
        const int row = it * RPI + lane / LPR;
        const u32x4_t v = *(const u32x4_t*)(stg + row * G::EPI_STRIDE + (lane % LPR) * 16);
        *(u32x4_t*)(ow + (size_t)row * D + (lane % LPR) * 8) = v;
      }
    }
    W4U_STAMP(10);  // O stores issued
#ifdef W4U_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    W4U_STAMP(11);  // ... and acknowledged
    if (blockIdx.x == 0 && wave == 0 && lane == 0) {
      *(volatile unsigned long long*)(smem + W4U<D>::LDS + 8 * 15) = __builtin_amdgcn_s_memrealtime();
      for (int i = 0; i < 16; ++i) g_w4u_stamps[i] = *(volatile unsigned long long*)(smem + W4U<D>::LDS + 8 * i);
    }
#endif
    if (!has_next) break;
    vb = vbn;
    bh = bhn;
    q0 = q0n;
  }
  if constexpr (WALK == 2) {
    // the last workgroup out zeroes the slot: every workgroup has made its final claim before it gets here
    if (wave == 0 && lane == 0) {
      if (atomicAdd(queue + 8, 1u) == (unsigned)nwg - 1u) {
#pragma unroll
        for (int i = 0; i < 9; ++i) __hip_atomic_store(queue + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

// ---- split-KV combine: O[row] = sum_s w_s O_s[row] / sum_s w_s, w_s = 2^(L_s[row] - max_s L_s[row]).  One thread per 8 output
// columns (16 B in per split, 16 B out); rows = B H N.  HBM-bound and tiny next to the attention itself: nsplit + 1 rows of D halves.
template <int D>
__global__ __launch_bounds__(256) void attn_split_combine_kernel(const half_t* __restrict__ Op, const float* __restrict__ lse,
                                                                 half_t* __restrict__ O, int nsplit, size_t rows) {
  constexpr int C8 = D / 8;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t row = idx / C8;
  const int c = (int)(idx % C8);
  if (row >= rows) return;
  float mx = lse[row];
  for (int s = 1; s < nsplit; ++s) mx = fmaxf(mx, lse[(size_t)s * rows + row]);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float den = 0.f;
  for (int s = 0; s < nsplit; ++s) {
    const float w = __builtin_amdgcn_exp2f(lse[(size_t)s * rows + row] - mx);
    const half8_t v = *(const half8_t*)(Op + ((size_t)s * rows + row) * D + 8 * c);
    den += w;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += w * (float)v[e];
  }
  const float inv = 1.0f / den;
  half8_t o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (half_t)(acc[e] * inv);
  *(half8_t*)(O + row * D + 8 * c) = o;
}

}  // namespace lc
