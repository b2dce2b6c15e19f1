/*
 * lc_abi.h — C-ABI of libleetcuda_amd.so (MI355X / gfx950 only).
 *
 * This is the drop-in boundary for the two hot paths of xlite-dev/LeetCUDA:
 *   kernels/hgemm      (fp16 GEMM,  C[M,N] = A[M,K] · B[K,N])
 *   kernels/flash-attn (FlashAttention-2 forward, O = softmax(QKᵀ/√D)·V)
 *
 * The reference binds these paths as flat lists of free functions
 * `void f(torch::Tensor ...)` registered by two pybind modules:
 *   kernels/hgemm/pybind/hgemm.cc:124-182            (38 exports, module `toy_hgemm` / `hgemm_lib`)
 *   kernels/flash-attn/pybind/flash_attn.cc:168-224  (26 exports + 3 optional, module `flash_attn_lib`)
 * Every one of those names is reachable through `lc_hgemm_call` / `lc_attn_call`
 * (dispatch by the reference's export name) and the typed entry points below.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch types; never throws across the boundary
 *   - all data pointers are DEVICE pointers (HBM); outputs are caller-allocated and written in place
 *     (same ownership rule as the reference wrappers, e.g. hgemm_mma_stage.cu:2331-2412)
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream, which is what the
 *     reference's `<<<grid, block, smem>>>` launches use)
 *   - return value: LC_OK (0) or a negative lc_status code; launches are asynchronous, launch
 *     failures surface as LC_ERR_LAUNCH, execution faults at the caller's next synchronize
 *     (same as the reference: no sync after launch)
 *   - there is NO CPU fallback anywhere behind this ABI.
 */
#ifndef LC_ABI_H_
#define LC_ABI_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LC_ABI_VERSION 2   /* 2: retired round-1 kernel variants, lc_build_info, lc_timer_*, probes moved to lc_diag.h */

typedef enum lc_status {
  LC_OK = 0,
  LC_ERR_ARG = -1,      /* null pointer, unknown enum / entry name                                  */
  LC_ERR_SHAPE = -2,    /* non-positive dims, shape / alignment the requested kernel family cannot tile  */
  LC_ERR_HEADDIM = -3,  /* head dim not supported by this attention family ("headdim not support!")  */
  LC_ERR_LAUNCH = -4,   /* hipLaunchKernel / hipFuncSetAttribute failed                              */
  LC_ERR_VENDOR = -5,   /* hipBLASLt comparator unavailable or failed                                */
  LC_ERR_DEVICE = -6    /* no gfx950 device visible                                                  */
} lc_status;

/* B operand storage.  NN: B is [K,N] row-major.  TN: B is stored [N,K] row-major (== column-major
 * [K,N]); the reference still presents it as a [K,N]-shaped tensor (kernels/hgemm/tools/utils.py:152-156)
 * and takes the layout from the entry-point NAME, never from strides — so does this ABI. */
typedef enum lc_layout { LC_LAYOUT_NN = 0, LC_LAYOUT_TN = 1 } lc_layout;

/* HGEMM kernel families behind the ABI (numeric values are stable; 2, 5, 7, 8, 11 were experiments that
 * measured slower and were retired — passing them returns LC_ERR_ARG). */
typedef enum lc_hgemm_variant {
  LC_HGEMM_AUTO = 0,       /* best available for the shape: the reference's legal shapes (M, N % 128 == 0, K % 32 == 0, K >= 64;
                              hgemm_mma_stage.cu:650,675-676) run MFMA256W4Y when the 256-tileable interior has > 128 tiles (128-wide
                              border strips on MFMA128 in a second launch), MID / MFMA128 otherwise; every other shape RAGGED, KPAD, EDGE or GENERIC */
  LC_HGEMM_MFMA256 = 1,    /* 256x256x64 WG tile, 8 wave64, LDS-DMA double buffer, one barrier / K-tile (simplest)      */
  LC_HGEMM_GENERIC = 3,    /* 64x64x32 edge-predicated MFMA kernel: any M,N,K                                           */
  LC_HGEMM_MFMA256P2 = 4,  /* 8-wave ping-pong, 2 phases of 16 MFMAs per K tile, DMA issued inside the MFMA clusters:
                              the independently scheduled cross-check of the default kernel                            */
  LC_HGEMM_MFMA128 = 6,    /* 128x128x64 tile, 4 wave64: M, N multiples of 128 (the reference's own tile), K of 32 (>= 64)  */
  LC_HGEMM_MFMA256W4B = 9, /* 256x256x64 tile, FOUR wave64 with 128x128 wave tiles, A ring of 2 + B ring of 3 K tiles,      */
                           /* v_mfma_f32_32x32x16_f16, global_load_lds DMA (64-bit addresses: the fallback for > 2 GiB spans) */
  LC_HGEMM_MFMA256W4C = 10, /* W4B with buffer_load ... lds (descriptor + scalar offset) DMA: the 32x32x16 baseline         */
  LC_HGEMM_MFMA256W4X = 12, /* W4C's ring / DMA schedule with v_mfma_f32_16x16x32_f16 (8 x 8 blocks of 16 x 16 per wave):   */
                            /* fewer joules per FLOP at the board power cap (hgemm_w4x.hip, compiler-scheduled; TN only)   */
  LC_HGEMM_MFMA256W4Y = 13, /* W4X with the K loop as one generated, hand-ordered instruction stream (hgemm_w4y.hip; TN and  */
                            /* NN): what LC_HGEMM_AUTO launches for large shapes.  Unlike the other 256-tile kernels (M, N % 256 == 0, */
                            /* K % 64 == 0) it takes M, N % 128 == 0 (>= 256) and K % 32 == 0 (>= 64): border strips + a half K-step  */
  LC_HGEMM_MID = 14,        /* the mid-size kernel (hgemm_mid.hip, round 6): (64 | 128) x (128 | 192) x 64 tile, 4 wave64, 2 - 3 slot LDS */
                            /* ring, hand-ordered asm K loop: LC_HGEMM_AUTO's choice between the eight-wave 128-tile kernel (small     */
                            /* grids) and the 256-tile kernel (> 128 tiles of 256 x 256) — n = 1280 .. 2816 square — with the tile that */
                            /* leaves the least work on the busiest CU.  M, N % 64 == 0 (a tile must divide them), K % 32 == 0 (>= 64)  */
  LC_HGEMM_EDGE = 15,       /* the vectorised edge kernel (hgemm_edge.hip, late round 6): 128 x 128 x 64 tile, any M and N, K % 8 == 0 (NN: */
                            /* N % 8 == 0), 16-byte chunks from clamped addresses: what LC_HGEMM_AUTO runs where neither a tiled kernel    */
                            /* nor LC_HGEMM_RAGGED applies (K % 32 != 0, K < 64); K % 8 != 0 / unaligned pointers stay on LC_HGEMM_GENERIC  */
  LC_HGEMM_RAGGED = 16,     /* ragged M / N with K % 32 == 0 (K >= 64), N % 8 == 0 (LC_HGEMM_AUTO's choice there): more than half a CU's worth of 256 x 256  */
                            /* tiles: the interior they divide on hgemm_w4y_kernel, the L-shaped border on hgemm_mid_edge_kernel (128 x 128 tiles of the     */
                            /* mid-size kernel that reach beyond M / N: clamped sources, predicated stores) in a second launch; else all of it on that kernel */
                            /* (tile: "hgemm_ragged_tile"; split-K with workspace partials as "hgemm_mid_splitk" says, one K range under graph capture)      */
  LC_HGEMM_KPAD = 17,       /* K % 32 != 0 (K % 8 == 0, N % 8 == 0, K >= 256) (LC_HGEMM_AUTO from a quarter of a 128 x 128 block per CU on, "hgemm_kpad"): A and B   */
                            /* copied into the stream's workspace with K zero-padded to a multiple of 32, then LC_HGEMM_AUTO on the padded problem (exactly the */
                            /* same result: zeros add nothing; the padded problem runs its workspace-free form — lc_hgemm_kernel_name reports its default launch, */
                            /* a split-K suffix " xN" there does not apply); hgemm_edge_kernel under graph capture / without workspace                          */
  /* the reference's "CUDA-core" ladder as vector-ALU kernels (hgemm_valu.hip; NN only; v_dot2c_f32_f16, fp32 accumulate);
   * shapes a rung does not tile (and TN) run LC_HGEMM_GENERIC */
  LC_HGEMM_VALU_NAIVE = 20,                  /* one thread per C element, operands from global memory                       */
  LC_HGEMM_VALU_SLICED_K = 21,               /* 32x32x32 LDS tile, one C element per thread                                 */
  LC_HGEMM_VALU_T8X8_X4 = 22,                /* 128x128 tile, 8x8 per thread, BK = 8, 8-byte global loads                   */
  LC_HGEMM_VALU_T8X8_X4_PACK = 23,           /* + vector LDS stores                                                         */
  LC_HGEMM_VALU_T8X8_X4_BCF = 24,            /* + bank-conflict-free LDS layout / thread tile                               */
  LC_HGEMM_VALU_T8X8_X4_PACK_BCF = 25,
  LC_HGEMM_VALU_T8X8_X8_PACK_BCF = 26,       /* 16-byte global loads                                                        */
  LC_HGEMM_VALU_T8X8_X8_PACK_BCF_DBUF = 27,  /* + double-buffered LDS                                                       */
  LC_HGEMM_VALU_T8X8_K16 = 28,               /* BK = 16                                                                     */
  LC_HGEMM_VALU_T8X8_K32 = 29,               /* BK = 32                                                                     */
  LC_HGEMM_VALU_T16X8_K32 = 30               /* 256x128 tile, 16x8 per thread                                               */
} lc_hgemm_variant;

/* FlashAttention-2 forward families: the reference's resource policies (kernels/flash-attn/mma/basic/ .cu files), kept as an enum
 * for signature parity ONLY.  `family`, `acc_f32` and `stages` of lc_attn_fwd_f16 are validated and then SELECT NOTHING: the kernel is
 * chosen from (D, N, V layout, B x H and the device's CU count) alone (lc_attn_kernel_name_bh reports it), every kernel accumulates
 * in fp32, and the LDS ring depth is fixed per kernel.  What a family name still decides is the head-dim limit of its ENTRY
 * (lc_attn_call / lc_attn_entry_info: the reference wrappers' switch(d)). */
typedef enum lc_attn_family {
  LC_ATTN_SPLIT_Q = 0,     /* flash_attn_mma_split_q.cu:55      K,V tiles staged in LDS, Q in regs   */
  LC_ATTN_SHARED_QKV = 1,  /* flash_attn_mma_share_qkv.cu:70    one LDS arena time-shared by K and V */
  LC_ATTN_SHARED_KV = 2,   /* flash_attn_mma_share_kv.cu:70                                          */
  LC_ATTN_TILING_QK = 3,   /* flash_attn_mma_tiling_qk.cu:76    Q,K streamed in d-slices             */
  LC_ATTN_TILING_QKV = 4,  /* flash_attn_mma_tiling_qkv.cu:75   Q,K,V streamed in d-slices (FFPA)    */
  LC_ATTN_SPLIT_KV = 5     /* flash_attn_mma_split_kv.cu:33                                          */
} lc_attn_family;

int lc_abi_version(void);
const char* lc_status_string(int status);
/* 0 when a gfx950 device is current, LC_ERR_DEVICE otherwise. Writes the CU count when non-NULL. */
int lc_device_check(int* num_cus);

/* Build identification: the compile flags of this library as one string (e.g. "gfx950 -O3 LC_DIAG=0"); *is_diag = 1
 * when it was built with LC_DIAG=1 (ablation / stamp instantiations compiled in — bench.py and the tests refuse such a
 * library, its diagnosis knobs can make results WRONG).  Never fails. */
const char* lc_build_info(int* is_diag);

/* Run-time selection knobs for A-B benches (not part of the reference surface; correctness never depends on them):
 *   "attn_nw"      attention kernel for D <= 128: 0 = auto; 513 / 515 / 517 = the merged-phase 4-wave kernel attn_fwd_w4u_kernel<D, VT, WALK>
 *                  (attn_w4u.hip: D = 64 / 128, N % 256 == 0, V as [B,H,N,D] or [B,H,D,N]) with WALK 0 (one 256-row query block per
 *                  workgroup), 1 (persistent workgroup per CU, static walk) or 2 (persistent, dynamic per-XCD block queue) — the three compute
 *                  the same bits; 512 = round 2's name for 513; 514 = the same design with every phase as ONE generated asm statement
 *                  (attn_w4i.hip: D = 32 / 64 / 96 / 128, V as [B,H,N,D]; the only merged-phase kernel for D = 96 / 32, where auto picks it);
 *                  8 / 4 / 2 = lock-step kernel with that many waves (any N % (32 x waves) == 0, every D <= 128).  Auto: 515 up to N = 4096
 *                  (D = 64: up to N = 8192), 513 beyond (D = 64 / 128, N % 256 == 0); 514 for D = 96 / 32; else the lock-step kernel
 *   "attn_walk"    block walk of the merged-phase kernel under "attn_nw" = 0: 0 = auto by N (above), 1 / 2 / 3 = WALK 0 / 1 / 2
 *   "attn_w4i_sched" schedule 0 / 1 (default) of attn_w4i's generated phase statements (tools/gen_attn_w4i.py; same bits, A/B knob)
 *   "attn_d1024"   D = 1024 pair kernel (attn_bigd4.hip): a batch of 8 LDS-DMA pieces is spread over this many eighths of a half-phase: 0 = default (8), 2 / 4 / 6
 *   "w4y_sched"    schedule 0..2 of LC_HGEMM_MFMA256W4Y's generated loop body (tools/gen_hgemm_w4y.py; same bits, A/B knob; default 2 since round 6)
 *   "hgemm_persist" 1 (default) = LC_HGEMM_MFMA256W4Y as one persistent workgroup per CU walking the C tiles, when their number is a
 *                  multiple of the CU count and larger (same bits as the one-tile launch, 0: + 0.2 % at the cap, + 0.7 % zero-filled)
 *   "hgemm_stagger" K-loop stagger of LC_HGEMM_MFMA256W4Y (hgemm_w4y.hip): the workgroup starts its K walk at tile ((index & mask)
 *                  * step) mod (K / 64) and wraps, index = cx * XCD + cm * tile row + cn * tile column.  0 = auto (cx = 1, mask 7, step
 *                  K / 64 / 8: the eight XCDs start an eighth of K apart), 1 << 27 = off, else cx | cm << 4 | cn << 8 | step << 12 |
 *                  mask << 20 with mask < 128 (bit 27 together with any other bit is refused).  Only the fp32 accumulation order changes:
 *                  results agree with the unstaggered walk to fp16 rounding
 *                  (the K = 128 fp8 kernel shares the stagger and the persistent walk of "hgemm_persist")
 *   "attn_split"   split-KV of the merged-phase kernel (attn_w4u.hip WALK 3) for grids that do not fill the GPU: 0 = auto (a cost model
 *                  over B x H, N, D and the CU count picks 1 / 2 / 4 / 8 / 16 KV ranges per 256-row query block, lc_abi.hip attn_split_auto),
 *                  1 = off: the merged-phase kernel itself on any grid (also switches off the small-grid substitution below), 2 / 4 / 8 / 16 = that
 *                  factor on any grid (N / 64 divisible by it, >= 2 tiles per range).  Partials live in a
 *                  cached per-stream workspace (a few MiB, never freed); while a stream is being captured the unsplit kernel runs
 *                  Small grids the rule leaves unsplit (<= half a GPU of 256-row blocks, N <= 2048) run the 4-wave lock-step kernel, whose
 *                  128-row workgroups fill twice the CUs (+ 9 ... 12 %)
 *   "attn_bigd_map" block -> query block map of the D = 1024 / D = 512 kernels: 1 = every XCD owns consecutive query blocks of a head (its 32
 *                  CUs share one pass over the head's K / V: the fewest fabric bytes), 2 = round-robin over the XCDs (every XCD streams every
 *                  head: ~2 x the fabric bytes, but the 8 XCDs walk the same heads out of the Infinity Cache); 0 = auto: 2 for D = 1024
 *                  (+ 3.7 %), 1 for D = 512 (2: - 2 %).  Same bits
 *   "attn_bigd_stagger" D = 1024 kernel: the workgroups of XCD x start their KV walk x eighths of the sequence in and wrap (only the fp32 summation
 *                  order changes): 0 = auto (on with the round-robin block map: + 2 %), 1 = off, 2 = on
 *   "hgemm_splitk" split-K of the 128-tile blocks that serve the border strips (M, N % 256 == 128) / the ragged last wave of
 *                  LC_HGEMM_MFMA256W4Y: 0 = auto (2 CUs' worth of blocks per tile when the launch holds fewer blocks than CUs, every K
 *                  range >= 8 tiles), 1 = off, 2 .. 8 = that factor; fp32 partials in the same workspace + a reduce kernel
 *   "hgemm_128w"   waves of LC_HGEMM_MFMA128: 0 = auto (eight — the two k-steps of every K tile on two groups of four waves, summed through LDS at the
 *                  end — on grids of <= 0.6 blocks per CU: + 11 % at 1024^3 / 1536^3; level at 2048^3, slower for NN beyond), 1 = four, 2 = eight
 *   "hgemm_mid"    LC_HGEMM_AUTO's use of LC_HGEMM_MID: 0 = auto (among the 64 / 128 / 192 x 128 / 192 tiles that divide the
 *                  problem the one with the least ceil(workgroups / CUs) x tile area), 1 = never, 12 / 13 / 22 / 23 / 32 / 33 = that tile (rows / 64,
 *                  columns / 64) whenever it divides the problem (A/B knob; also what an explicit LC_HGEMM_MID then runs)
 *   "hgemm_mid_ns" LDS ring slots of LC_HGEMM_MID: 0 = auto (3 when the grid is one round of <= one workgroup per CU, else 2), 2, 3
 *   "hgemm_mid_splitk"  split-K of LC_HGEMM_MID (64 / 128 x 128 tiles; fp32 partials in the stream's workspace + a reduce launch; none under graph
 *                  capture): 0 = auto (a one-round grid on <= half the CUs with >= 64 K tiles: as many ranges as fill the CUs, >= 32 K tiles each,
 *                  <= 8), 1 = never, 2 .. 8 = that many ranges wherever legal (A/B)
 *   "hgemm_tail"   (2 = as 1, but the quadrants on the 128-tile kernel with a workspace split-K: round 5's form, kept for A/B and for shapes with
 *                  border strips; 3 / 4 = as 1 with remainders up to 0.75 / 1.0 of the CUs: measured 8 ... 18 % slower, A/B only)  1 (default) = when the 256-tile grid's last wave holds at most 128 tiles, the generated-loop kernel computes the
 *                  full waves and 128 x 128 blocks the four quadrants of each remaining tile (round 6: on the mid-size kernel, + 4 ... 6 % at 4352 ... 6400); 0 = one launch
 *   "hgemm_tail_tile"  sub-tiles of that tail on the mid-size kernel: 0 = auto (64 x 128 eighths while 8 x the remaining tiles fit one round of the
 *                  CUs, else 128 x 128 quadrants), 1 = eighths, 2 = quadrants (bit-identical results; A/B knob)
 *   "hgemm_ragged" LC_HGEMM_AUTO on ragged M / N with K % 32 == 0 (K >= 64), N % 8 == 0: 0 = LC_HGEMM_RAGGED (the tiled kernels; 128 x 128 tiles with clamped
 *                  sources and predicated stores on what they do not divide), 1 = never (hgemm_edge_kernel alone; A/B knob)
 *   "hgemm_kpad"   LC_HGEMM_AUTO on K % 32 != 0 (K % 8 == 0, N % 8 == 0): 0 = auto (LC_HGEMM_KPAD once the shape holds a quarter of a 128 x 128 block per CU), 1 = never
 *                  (hgemm_edge_kernel), 2 = wherever legal (A/B knob)
 *   "hgemm_ragged_tile"  tile of a ragged problem that runs entirely on hgemm_mid_edge_kernel: 0 = auto (the smallest of 64 x 128, 128 x 128, 128 x 192 (TN) /
 *                  192 x 128 (NN), 192 x 192 (TN) whose grid fits one round of the CUs, three ring slots; else 128 x 128 with two), 12 / 22 / 23 / 32 / 33 = that
 *                  tile where the layout has it (A/B knob)
 *   "hgemm_ragged_fork"  LC_HGEMM_RAGGED's border launch on a per-device side stream forked from / joined to the caller's stream by events (runs beside
 *                  the interior; never while the caller's stream is being captured): 0 = auto (only beside an unsplit last round of the 256-tile grid that
 *                  leaves >= 3 / 8 of the CUs idle: + 4 %; beside full rounds it costs 2 ... 11 %), 1 = never, 2 = always (same bits)
 *   "hgemm_raster" block -> C tile map of the tiled GEMM kernels: 0 = auto (2 when A + B exceed 272 MiB — the 256 MiB Infinity Cache and a margin,
 *                  else 1), 1 = the reference's block swizzle (N panels of swizzle_stride columns, every XCD a contiguous id
 *                  range), 2 = XCD super-block raster (16 x 16 tile steps shared through the Infinity Cache, 4 x 8 per XCD;
 *                  swizzle_stride ignored)
 *   "hgemm_auto"   kernel LC_HGEMM_AUTO launches on large 256-tileable shapes (a 256-tile lc_hgemm_variant value)
 *   "fp8_mx"       fp8 GEMM (lc_gemm_fp8_e4m3): 3 = MX-scaled K = 128 MFMA (v_mfma_scale_f32_16x16x128_f8f6f4), generated loop (default);
 *                  1 = MX K = 64 MFMA, compiler-scheduled 4-wave kernel (the cross-check); 2 = MX K = 64, 8-wave kernel; 0 = plain K = 16
 *   "attn_d512"    D = 256 / 512 / 1024 kernel: 0 = auto (D = 256, N % 256 == 0, either V layout: 64 query rows per wave on v_mfma_f32_16x16x32 with K / V
 *                  rings, attn_bigd7.hip; D = 512: the full-width kernel on the same MFMA shape, attn_bigd6.hip; D = 256 with
 *                  N % 256 == 128: attn_bigd2.hip; D = 1024: two waves share 32 query rows and split the head dim, attn_bigd4.hip),
 *                  1 = round-1 column-split kernel, 2 = 32-row double-buffered tiles (attn_bigd3.hip: validated on hardware in round 3,
 *                  6-8 % slower, a cross-check), 3 = D = 256 / 512 on the other MFMA shape than auto (attn_bigd2.hip: v_mfma_f32_32x32x16),
 *                  4 = auto, but attn_bigd7 also on grids that do not fill the GPU (auto hands D = 256 launches of fewer than about 0.75
 *                  workgroups of 256 query rows per CU to attn_bigd2.hip, whose workgroups own 128 rows; lc_attn_kernel_name reports the
 *                  kernel of a grid that fills the GPU, lc_attn_kernel_name_bh the one a launch of BH problems runs)
 * Diagnosis keys (include/lc_diag.h) are rejected with LC_ERR_ARG unless the library was built with LC_DIAG=1. */
int lc_tune_set(const char* key, int value);
/* Current and default value of a knob (either pointer may be NULL); LC_ERR_ARG for an unknown key.  lc_tune_count / lc_tune_key
 * enumerate the keys (NULL for a diagnosis key in a production library).  The knobs are process-wide atomics: setting one while
 * another host thread launches is well defined (that launch sees the old or the new value), but A/B benches should not rely on it. */
int lc_tune_get(const char* key, int* value, int* default_value);
int lc_tune_count(void);
const char* lc_tune_key(int index);
/* Measures the constants of the split-KV launch rule on the CURRENT device (microseconds per KV tile at D = 128 / 64, fixed cost of the
 * combine launch, bytes per microsecond of partial traffic: about 60 tiny launches on zero-filled scratch, a few milliseconds) and makes
 * the rule use them for this device from now on; the built-in constants (fitted on the round-5 boxes) stay when this is never called or when
 * the measurement lies outside [0.4, 2.5] x of them (LC_ERR_ARG; out4 still receives what was measured).  out4 (optional): tau128, tau64,
 * x0_us, bytes_per_us.  bench.py calls it once per process; a caller that never does gets round 5's behaviour.  Not while `stream` is
 * being captured.  Knobs: "attn_calib" = 1 ignores the measurement; "rule_cus" = n makes every launch RULE (not the grids) reason with n
 * CUs instead of the device's own — for tests of the rules (64 .. 1024; 0 = the device). */
int lc_tune_calibrate(void* stream, float* out4);

/* Internal workspace (split-KV attention partials, split-K border strips of the GEMM): one cached device buffer per
 * (device, stream), grown geometrically, at most 16 per device and 1 GiB each; requests beyond that and launches on a stream that is
 * being captured run the workspace-free form of the same computation.  lc_workspace_bytes: bytes currently cached over all devices;
 * lc_workspace_release: hipFree every cached buffer (waits for the devices), returns the bytes given back — for callers that want
 * the memory back after a phase of small-grid launches.  Neither has a reference counterpart (its kernels allocate nothing). */
size_t lc_workspace_release(void);
size_t lc_workspace_bytes(void);

/* ---- HGEMM ------------------------------------------------------------------------------------
 * Replaces the host launchers + kernels of kernels/hgemm/mma/basic/hgemm_mma_stage.cu:644-1052,2284-2412
 * (NN) and kernels/hgemm/mma/swizzle/hgemm_mma_stage_tn_swizzle_x4.cu:207,892 (TN).
 * fp16 in, fp32 MFMA accumulate, fp16 out; no alpha/beta.
 * `stages` (2..5 in the reference, anything else falls back to 2: hgemm_mma_stage.cu:2388-2391) is ACCEPTED AND
 * IGNORED: the LDS ring depth is a fixed property of each kernel family here (2-slot rings; 2 + 3 slots for the W4
 * kernels that fill the CU's 160 KiB) — results never depend on it, exactly as in the reference.
 * `swizzle_stride` is the reference's thread-block-swizzle N-panel width in columns (hgemm.py:198-208): <= 1 means no
 * panel rasterisation.  Shapes: the tuned variants need M, N multiples of 256 (128 for MFMA128), K a multiple of 64
 * and 16-byte aligned pointers, else LC_ERR_SHAPE; LC_HGEMM_AUTO and LC_HGEMM_GENERIC accept ANY positive M, N, K and
 * 2-byte alignment (the edge-predicated kernel serves what the tiles do not divide; all address arithmetic is 64-bit). */
int lc_hgemm_f16(const void* A, const void* B, void* C, int M, int N, int K, int layout,
                 int variant, int stages, int swizzle_stride, void* stream);

/* Vendor comparator = the reference's cuBLAS entry points (kernels/hgemm/cublas/hgemm_cublas.cu:15-68,
 * 196-229) on hipBLASLt: init/destroy of a process-global handle + NN/TN GEMM, fp32 compute. */
int lc_vendor_init(void);
int lc_vendor_destroy(void);
int lc_hgemm_vendor_f16(const void* A, const void* B, void* C, int M, int N, int K, int layout,
                        void* stream);

/* EXTENSION (BASELINE config 5, no reference counterpart — the reference wrappers hard-require kHalf):
 * fp8 GEMM, OCP e4m3fn inputs, fp32 MFMA accumulate, fp16 output:  C[M,N] = alpha * A8[M,K] * B8^T with B8
 * stored [N,K] (the TN layout).  M, N multiples of 256, K multiple of 128, 16-byte aligned pointers. */
int lc_gemm_fp8_e4m3(const void* A, const void* B, void* C, int M, int N, int K, float alpha,
                     int swizzle_stride, void* stream);

/* EXTENSION: the same GEMM on OCP MX data — every 32 consecutive k of every row of A and of B carry an E8M0 block scale
 * (value = e4m3 * 2^(scale - 127); scale 127 = 1.0, 255 = NaN per the MX spec):
 *   C[m][n] = alpha * sum_k A8[m][k] 2^(SA[m][k/32] - 127) * B8[n][k] 2^(SB[n][k/32] - 127)
 * The matrix core wants one scale byte per lane (row, k block) and instruction; lc_mxfp8_pack_scales reorders the natural
 * [rows][K/32] byte array S into the dword array P (rows * K / 32 bytes, 8-byte aligned) that lc_gemm_mxfp8 consumes — once per weight
 * matrix, per call for activations.  rows multiple of 256, K multiple of 128.  P's layout is an implementation detail
 * (leetcuda_amd/csrc/gemm_fp8_w4k.hip); treat it as opaque. */
int lc_mxfp8_pack_scales(const void* S, void* P, int rows, int K, void* stream);
int lc_gemm_mxfp8(const void* A, const void* PA, const void* B, const void* PB, void* C, int M, int N, int K, float alpha,
                  int swizzle_stride, void* stream);

/* Dispatch by the reference's export name (hgemm.cc:126-181). 3-argument entries ignore
 * stages/swizzle/swizzle_stride. `init_cublas_handle` / `destroy_cublas_handle` take no tensors
 * (pass NULLs and zeros). */
int lc_hgemm_call(const char* entry, const void* A, const void* B, void* C, int M, int N, int K,
                  int stages, int swizzle, int swizzle_stride, void* stream);
int lc_hgemm_entry_count(void);
const char* lc_hgemm_entry_name(int index);
/* layout (lc_layout), number of reference arguments (0, 3 or 6); returns LC_ERR_ARG if unknown. */
int lc_hgemm_entry_info(const char* entry, int* layout, int* nargs);

/* ---- FlashAttention-2 forward -------------------------------------------------------------------
 * Replaces kernels/flash-attn/mma/basic/flash_attn_mma_split_q.cu:55,702,769,
 * flash_attn_mma_share_qkv.cu:70,772,872, flash_attn_mma_tiling_qkv.cu:75,800,881 and siblings.
 * Q,K,O: [B,H,N,D] fp16 contiguous.  V: [B,H,N,D], or [B,H,D,N] when v_transposed != 0
 * (the reference's *_swizzle_qkv share_kv/share_qkv/tiling_qk entries, flash_attn_mma.py:441-442).
 * Non-causal, scale = 1/sqrt(D), no dropout / mask / LSE output.  fp32 softmax, fp32 MFMA accumulate
 * (family / acc_f32 / stages are accepted for signature parity and select nothing; CDNA4 MFMA has no fp16-accumulate form).
 * Batch variance: the kernel — and with it the fp32 summation order, i.e. the low bits of O — depends on B x H and the CU count
 * for shapes whose grid would not fill the GPU (small grids run 128-row workgroups or the split-KV path); one (B, H, N, D) on one
 * device is bit-reproducible from launch to launch.
 * N must be a multiple of 64; D in {32, 64, 96, 128, 256, 512, 1024} — the head dims of the reference dispatchers. */
int lc_attn_fwd_f16(const void* Q, const void* K, const void* V, void* O, int B, int H, int N, int D,
                    int v_transposed, int family, int acc_f32, int stages, void* stream);

/* EXTENSION (BASELINE config 5 "FFPA-style QKV fine-grained tiling D=512 bf16"; the reference has no bf16
 * entry): the large-head-dim d-slice tiling kernel on bfloat16 Q,K,V,O [B,H,N,D], D in {256, 512}. */
int lc_attn_fwd_bf16(const void* Q, const void* K, const void* V, void* O, int B, int H, int N, int D,
                     void* stream);

/* Dispatch by the reference's export name (flash_attn.cc:170-223). */
int lc_attn_call(const char* entry, const void* Q, const void* K, const void* V, void* O, int B, int H,
                 int N, int D, int stages, void* stream);
int lc_attn_entry_count(void);
const char* lc_attn_entry_name(int index);
/* family (lc_attn_family), v_transposed, acc_f32, max head dim for stages>1 / stages<=1 and number of
 * reference arguments (4 for flash_attn_cute, else 5). */
int lc_attn_entry_info(const char* entry, int* family, int* v_transposed, int* acc_f32,
                       int* max_d_stage2, int* max_d_stage1, int* nargs);

/* ---- introspection (bench.py labels its roofline / rocprof rows with these; never launches) ------
 * Name of the kernel the dispatcher would launch for this call with the current lc_tune_set selection, in the form
 * rocprofv3 demangles it to (e.g. "hgemm_w4b_kernel<false,true,false,0>"), written NUL-terminated into buf.
 * Pointers are assumed 16-byte aligned.  Returns LC_OK, LC_ERR_ARG / LC_ERR_SHAPE / LC_ERR_HEADDIM as the call would. */
int lc_hgemm_kernel_name(int M, int N, int K, int layout, int variant, char* buf, int buflen);
int lc_attn_kernel_name(int N, int D, int v_transposed, int bf16, char* buf, int buflen);
/* the same for a launch of BH = batch x heads problems: where the choice depends on how far the grid fills the GPU (D = 256, "attn_d512"
 * above) the name is the kernel THAT launch runs; BH <= 0 = lc_attn_kernel_name (a grid that fills the GPU) */
int lc_attn_kernel_name_bh(int BH, int N, int D, int v_transposed, int bf16, char* buf, int buflen);
/* How often the overflow slow path of the merged-phase attention kernels (attn_w4u.hip, attn_w4i.hip) ran since the last reset:
 * out4 = { executions, sum of their KV half-tile indices, executions that saw a non-finite row sum, bit pattern (fp32) of the
 * last offending row sum }.  Synchronises the device (hipMemcpyFromSymbol).  out4 may be NULL (reset only). */
int lc_attn_slowpath_stats(unsigned* out4, int reset);

/* ---- measurement helpers (used by bench.py; HIP events on the launch stream) --------------------
 * Launch `iters` back-to-back calls between two hipEvents recorded on `stream`; returns average
 * milliseconds per launch in *ms_per_launch.  Any HIP failure (event, launch, execution fault) -> LC_ERR_LAUNCH. */
int lc_hgemm_time(const void* A, const void* B, void* C, int M, int N, int K, int layout, int variant,
                  int stages, int swizzle_stride, int warmup, int iters, void* stream,
                  float* ms_per_launch);
int lc_attn_time(const void* Q, const void* K, const void* V, void* O, int B, int H, int N, int D,
                 int v_transposed, int family, int stages, int warmup, int iters, void* stream,
                 float* ms_per_launch);
/* Generic form: bracket ANY sequence of launches on `stream` with HIP events.  lc_timer_start records the first
 * event and returns an opaque timer; lc_timer_stop records the second, waits for it, writes the elapsed
 * milliseconds and frees the timer (also on failure). */
int lc_timer_start(void* stream, void** timer);
int lc_timer_stop(void* timer, float* elapsed_ms);
/* Effective shader clock without a profiler: a one-wave kernel writes {s_memtime (shader cycles), s_memrealtime
 * (constant 100 MHz)} into out_u64x2 (device memory).  Two probes around a timed batch on the same stream give
 * eff_clock = d(memtime) / (d(memrealtime) / 100e6). */
int lc_clock_probe(void* out_u64x2, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LC_ABI_H_ */
