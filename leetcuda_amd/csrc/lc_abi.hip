// lc_abi.hip — the C-ABI of libleetcuda_amd.so (declared in include/lc_abi.h): argument checks,
// kernel selection, launch geometry, the reference entry-name tables and HIP-event timing helpers.
// Host side of the reference's L2 layer (the `void f(torch::Tensor...)` wrappers at the tail of every
// reference .cu, e.g. kernels/hgemm/mma/basic/hgemm_mma_stage.cu:2331-2412 and
// kernels/flash-attn/mma/basic/flash_attn_mma_split_q.cu:701-815) re-expressed on raw pointers.
#include "../../include/lc_abi.h"

#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <new>

#include <mutex>
#include <utility>
#include <vector>

#include "lc_launch.h"
#include "attn_fwd.hip"
#include "hgemm_generic.hip"
#include "hgemm_edge.hip"
#include "hgemm_mfma128.hip"
#include "hgemm_mfma256.hip"
#include "hgemm_pingpong.hip"

using namespace lc;

namespace lc {
tune_t g_tune_attn_ablate{0};      // attention ablation / stamp builds (diagnosis only, LC_DIAG)
tune_t g_tune_w4_abl{0};           // hgemm_w4 ablation bits (diagnosis only, LC_DIAG)
tune_t g_tune_w4y_sched{2};        // hgemm_w4y_kernel loop schedule 0..2 (lc_tune_set "w4y_sched"; same bits; 2 since round 6: + 0.4 ... 4 % at 8704 ... 12800, level at 8192, profiles/r6i_hgemm_knob_sched_ab.log)
tune_t g_tune_hgemm_stamps{0};     // GEMM cycle-stamp builds (diagnosis only, LC_DIAG)
tune_t g_tune_hgemm_persist{1};    // 1 (default) = hgemm_w4y_kernel as a persistent workgroup per CU when the tiles divide evenly (lc_tune_set "hgemm_persist")
tune_t g_tune_hgemm_stagger{0};    // K-loop stagger of hgemm_w4y_kernel (lc_tune_set "hgemm_stagger"): 0 = auto (by XCD), 1 << 27 = off, else cx | cm << 4 | cn << 8 | step << 12 | mask << 20
tune_t g_tune_attn_bigd_stagger{0};   // attn_bigd4 (D = 1024): the KV walk of the workgroups on XCD x starts x eighths of the sequence in: 0 = auto (with the round-robin block map), 1 = off, 2 = on (results agree to rounding)
tune_t g_tune_attn_bigd_map{0};    // block -> query block map of attn_bigd4 / attn_bigd6: 0 = auto (D = 1024: round-robin over the XCDs, D = 512: XCD-contiguous), 1 = XCD-contiguous, 2 = round-robin (same bits; profiles/r5f_bigd_map.log)
tune_t g_tune_attn_d512{0};        // D = 256 / 512 / 1024: 0 = auto, 1 = column-split kernel, 2 = attn_bigd3, 3 = D = 256 / 512 on the other MFMA shape than auto (attn_bigd2 <-> attn_bigd7 / attn_bigd6), 4 = auto but attn_bigd7 on any grid
}  // namespace lc

namespace {

// run-time tuning knobs (lc_tune_set): experiments and A/B benches, never required for correctness
tune_t g_tune_fp8_mx{3};                       // fp8 GEMM: 3 = MX K=128 MFMA, generated loop (gemm_fp8_w4k.hip); 1 = MX K=64, 4-wave kernel; 2 = MX K=64, 8-wave kernel; 0 = plain K=16 MFMA
tune_t g_tune_attn_w4i_sched{1};              // schedule of attn_fwd_w4i_kernel's generated phase statements (tools/gen_attn_w4i.py NSCHED; same bits)
tune_t g_tune_attn_nw{0};                    // attention kernel for D <= 128: 0 = auto, 513 / 515 / 517 / 514 / 8 / 4 / 2 (choose_attn_nw, lc_abi.h)
tune_t g_tune_attn_d1024{0};                 // attn_bigd4's DMA spread in eighths of a phase: 0 = default (8), 2 / 4 / 6 (A/B knob)
tune_t g_tune_attn_walk{0};                  // block walk of the merged-phase kernel under attn_nw = 0: 0 = auto by N, 1 / 2 / 3 = WALK 0 / 1 / 2
tune_t g_tune_attn_split{0};                 // split-KV of the merged-phase kernel on grids that do not fill the GPU: 0 = auto (attn_split_auto), 1 = off, 2 / 4 / 8 / 16 = that many KV ranges per query block
tune_t g_tune_hgemm_auto{LC_HGEMM_MFMA256W4Y};   // what LC_HGEMM_AUTO launches for large 256-tileable shapes (lc_tune_set "hgemm_auto")
tune_t g_tune_hgemm_splitk{0};                 // split-K of the 128-tile blocks that serve border strips / the ragged last wave: 0 = auto (launch_mfma256), 1 = off, 2 .. 8 = that factor
tune_t g_tune_rule_cus{0};                     // CU count the LAUNCH RULES reason with: 0 = the current device's own; 64 .. 1024 = that many (tests of the rules for other devices; grids are always sized with the real count)
tune_t g_tune_attn_calib{0};                   // split-KV cost model: 0 = the constants lc_tune_calibrate measured on this device when it ran (else the built-in ones), 1 = always the built-in ones
tune_t g_tune_hgemm_mid{0};                    // mid-size kernel (hgemm_mid.hip): 0 = auto (mid_tile_auto), 1 = never, 12 / 13 / 22 / 23 / 32 / 33 = that tile (rows / 64, columns / 64)
tune_t g_tune_hgemm_mid_ns{0};                 // ... its LDS ring slots: 0 = auto (3 for one-round grids, else 2), 2, 3
tune_t g_tune_hgemm_128w{0};                   // waves of the 128-tile kernel: 0 = auto (eight — intra-workgroup split-K — on grids of <= 0.6 blocks per CU), 1 = always four, 2 = always eight
tune_t g_tune_hgemm_tail{1};                   // 1 = hand the ragged last wave of the 256-tile kernel to 128 x 128 blocks (launch_mfma256: the mid-size kernel; 2 = round 5's 128-tile kernel + split-K), 0 = one launch
tune_t g_tune_hgemm_ragged{0};                 // LC_HGEMM_AUTO on ragged M / N with K % 32 == 0: 0 = LC_HGEMM_RAGGED (the tiled kernels, clamped 128 x 128 tiles on what they do not divide), 1 = never (hgemm_edge_kernel)
tune_t g_tune_hgemm_kpad{0};                   // LC_HGEMM_AUTO on K % 32 != 0 (K % 8 == 0): 0 = auto (zero-padded operand copies + the tuned kernels from a quarter of a 128 x 128 block per CU on), 1 = never (hgemm_edge_kernel), 2 = wherever legal
tune_t g_tune_hgemm_ragged_tile{0};            // tile of a ragged problem that runs entirely on hgemm_mid_edge_kernel: 0 = auto (ragged_plan), 12 / 22 / 23 / 32 / 33 = that tile (rows / 64, columns / 64; A/B)
tune_t g_tune_hgemm_ragged_fork{0};            // LC_HGEMM_RAGGED's border launch on a side stream, forked from and joined to the caller's (runs beside the interior): 0 = auto (launch_ragged), 1 = never, 2 = always
tune_t g_tune_hgemm_tail_tile{0};              // sub-tiles of the ragged tail on the mid-size kernel: 0 = auto (launch_mfma256), 1 = 64 x 128 eighths, 2 = 128 x 128 quadrants
tune_t g_tune_hgemm_mid_splitk{0};             // split-K of the mid-size kernel: 0 = auto (mid_tile_auto), 1 = never, 2 .. 8 = that many K ranges wherever legal (A/B)
tune_t g_tune_hgemm_raster{0};                 // block -> C tile map: 0 = auto (by operand footprint, panel_tiles), 1 = the reference's block swizzle (N panels from
                                             // swizzle_stride, XCD-contiguous ids), 2 = XCD super-block raster (hgemm_mfma256.hip raster_xcd16)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ------------------------------------------------------------------------------------------------
// reference entry tables
struct HgemmEntry {
  const char* name;
  int layout;   // lc_layout
  int nargs;    // 0, 3 or 6
  int variant;  // lc_hgemm_variant, -1 = vendor, -2 = handle init, -3 = handle destroy
};

#define NN LC_LAYOUT_NN
#define TN LC_LAYOUT_TN
// kernels/hgemm/pybind/hgemm.cc:126-181, in the reference's registration order.
const HgemmEntry kHgemmEntries[] = {
    {"hgemm_naive_f16", NN, 3, LC_HGEMM_VALU_NAIVE},
    {"hgemm_sliced_k_f16", NN, 3, LC_HGEMM_VALU_SLICED_K},
    {"hgemm_t_8x8_sliced_k_f16x4", NN, 3, LC_HGEMM_VALU_T8X8_X4},
    {"hgemm_t_8x8_sliced_k_f16x4_pack", NN, 3, LC_HGEMM_VALU_T8X8_X4_PACK},
    {"hgemm_t_8x8_sliced_k_f16x4_bcf", NN, 3, LC_HGEMM_VALU_T8X8_X4_BCF},
    {"hgemm_t_8x8_sliced_k_f16x4_pack_bcf", NN, 3, LC_HGEMM_VALU_T8X8_X4_PACK_BCF},
    {"hgemm_t_8x8_sliced_k_f16x8_pack_bcf", NN, 3, LC_HGEMM_VALU_T8X8_X8_PACK_BCF},
    {"hgemm_t_8x8_sliced_k_f16x8_pack_bcf_dbuf", NN, 3, LC_HGEMM_VALU_T8X8_X8_PACK_BCF_DBUF},
    {"hgemm_t_8x8_sliced_k16_f16x8_pack_dbuf", NN, 3, LC_HGEMM_VALU_T8X8_K16},
    {"hgemm_t_8x8_sliced_k16_f16x8_pack_dbuf_async", NN, 3, LC_HGEMM_VALU_T8X8_K16},
    {"hgemm_t_8x8_sliced_k32_f16x8_pack_dbuf", NN, 3, LC_HGEMM_VALU_T8X8_K32},
    {"hgemm_t_8x8_sliced_k32_f16x8_pack_dbuf_async", NN, 3, LC_HGEMM_VALU_T8X8_K32},
    {"hgemm_t_16x8_sliced_k32_f16x8_pack_dbuf", NN, 3, LC_HGEMM_VALU_T16X8_K32},
    {"hgemm_t_16x8_sliced_k32_f16x8_pack_dbuf_async", NN, 3, LC_HGEMM_VALU_T16X8_K32},
    {"init_cublas_handle", NN, 0, -2},
    {"destroy_cublas_handle", NN, 0, -3},
    {"hgemm_cublas_tensor_op_nn", NN, 3, -1},
    {"hgemm_cublas_tensor_op_tn", TN, 3, -1},
    {"hgemm_wmma_m16n16k16_naive", NN, 3, LC_HGEMM_GENERIC},
    {"hgemm_wmma_m16n16k16_mma4x2", NN, 3, LC_HGEMM_GENERIC},
    {"hgemm_wmma_m16n16k16_mma4x2_warp2x4", NN, 3, LC_HGEMM_MFMA128},
    {"hgemm_wmma_m16n16k16_mma4x2_warp2x4_dbuf_async", NN, 3, LC_HGEMM_MFMA128},
    {"hgemm_wmma_m32n8k16_mma2x4_warp2x4_dbuf_async", NN, 3, LC_HGEMM_MFMA128},
    {"hgemm_wmma_m16n16k16_mma4x2_warp2x4_stages", NN, 6, LC_HGEMM_MFMA256},
    {"hgemm_wmma_m16n16k16_mma4x2_warp2x4_stages_dsmem", NN, 6, LC_HGEMM_MFMA256},
    {"hgemm_wmma_m16n16k16_mma4x2_warp4x4_stages_dsmem", NN, 6, LC_HGEMM_MFMA256P2},
    {"hgemm_wmma_m16n16k16_mma4x4_warp4x4_stages_dsmem", NN, 6, LC_HGEMM_MFMA256P2},
    {"hgemm_mma_m16n8k16_naive", NN, 3, LC_HGEMM_GENERIC},
    {"hgemm_mma_m16n8k16_mma2x4_warp4x4", NN, 3, LC_HGEMM_MFMA256},
    {"hgemm_mma_m16n8k16_mma2x4_warp4x4_stages", NN, 6, LC_HGEMM_MFMA256},
    {"hgemm_mma_m16n8k16_mma2x4_warp4x4_stages_dsmem", NN, 6, LC_HGEMM_MFMA256},
    {"hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem", NN, 6, LC_HGEMM_AUTO},
    {"hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_x4", NN, 6, LC_HGEMM_AUTO},
    {"hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_rr", NN, 6, LC_HGEMM_AUTO},
    {"hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_swizzle", NN, 6, LC_HGEMM_AUTO},
    {"hgemm_mma_m16n8k16_mma2x4_warp4x4_stages_dsmem_tn", TN, 6, LC_HGEMM_MFMA256},
    {"hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_tn_swizzle_x4", TN, 6, LC_HGEMM_AUTO},
    {"hgemm_mma_stages_block_swizzle_tn_cute", TN, 6, LC_HGEMM_AUTO},
};
#undef NN
#undef TN
constexpr int kNumHgemmEntries = sizeof(kHgemmEntries) / sizeof(kHgemmEntries[0]);

struct AttnEntry {
  const char* name;
  int family, vt, acc_f32, maxd_s2, maxd_s1, nargs;
};
// kernels/flash-attn/pybind/flash_attn.cc:170-223; head-dim limits from each wrapper's switch(d)
// (e.g. flash_attn_mma_split_q.cu:769-815, flash_attn_mma_share_qkv.cu:872-921).
const AttnEntry kAttnEntries[] = {
    {"flash_attn_mma_stages_split_kv", LC_ATTN_SPLIT_KV, 0, 0, 128, 128, 5},
    {"flash_attn_mma_stages_split_q", LC_ATTN_SPLIT_Q, 0, 0, 128, 128, 5},
    {"flash_attn_mma_stages_split_q_shared_kv", LC_ATTN_SHARED_KV, 0, 0, 128, 256, 5},
    {"flash_attn_mma_stages_split_q_shared_qkv", LC_ATTN_SHARED_QKV, 0, 0, 128, 256, 5},
    {"flash_attn_mma_stages_split_q_tiling_qk", LC_ATTN_TILING_QK, 0, 0, 1024, 1024, 5},
    {"flash_attn_mma_stages_split_q_tiling_qkv", LC_ATTN_TILING_QKV, 0, 0, 1024, 1024, 5},
    {"flash_attn_mma_stages_split_q_shared_kv_acc_f32", LC_ATTN_SHARED_KV, 0, 1, 128, 256, 5},
    {"flash_attn_mma_stages_split_q_shared_qkv_acc_f32", LC_ATTN_SHARED_QKV, 0, 1, 128, 256, 5},
    {"flash_attn_mma_stages_split_q_tiling_qk_acc_f32", LC_ATTN_TILING_QK, 0, 1, 1024, 1024, 5},
    {"flash_attn_mma_stages_split_q_tiling_qkv_acc_f32", LC_ATTN_TILING_QKV, 0, 1, 1024, 1024, 5},
    {"flash_attn_mma_stages_split_q_shared_kv_swizzle_q", LC_ATTN_SHARED_KV, 0, 0, 128, 256, 5},
    {"flash_attn_mma_stages_split_q_shared_kv_swizzle_qk", LC_ATTN_SHARED_KV, 0, 0, 128, 256, 5},
    {"flash_attn_mma_stages_split_q_shared_kv_swizzle_qkv", LC_ATTN_SHARED_KV, 1, 0, 128, 256, 5},
    {"flash_attn_mma_stages_split_q_shared_qkv_swizzle_q", LC_ATTN_SHARED_QKV, 0, 0, 128, 256, 5},
    {"flash_attn_mma_stages_split_q_shared_qkv_swizzle_qk", LC_ATTN_SHARED_QKV, 0, 0, 128, 256, 5},
    {"flash_attn_mma_stages_split_q_shared_qkv_swizzle_qkv", LC_ATTN_SHARED_QKV, 1, 0, 128, 256, 5},
    {"flash_attn_mma_stages_split_q_tiling_qk_swizzle_q", LC_ATTN_TILING_QK, 0, 0, 1024, 1024, 5},
    {"flash_attn_mma_stages_split_q_tiling_qk_swizzle_qk", LC_ATTN_TILING_QK, 0, 0, 1024, 1024, 5},
    {"flash_attn_mma_stages_split_q_tiling_qk_swizzle_qkv", LC_ATTN_TILING_QK, 1, 0, 256, 256, 5},
    {"flash_attn_mma_stages_split_q_tiling_qkv_swizzle_q", LC_ATTN_TILING_QKV, 0, 0, 1024, 1024, 5},
    {"flash_attn_mma_stages_split_q_tiling_qkv_swizzle_qk", LC_ATTN_TILING_QKV, 0, 0, 1024, 1024, 5},
    {"flash_attn_mma_stages_split_q_tiling_qkv_swizzle_qkv", LC_ATTN_TILING_QKV, 0, 0, 1024, 1024, 5},
    {"flash_attn_mma_stages_split_q_tiling_qkv_acc_f32_swizzle_q", LC_ATTN_TILING_QKV, 0, 1, 1024, 1024, 5},
    {"flash_attn_mma_stages_split_q_tiling_qkv_acc_f32_swizzle_qk", LC_ATTN_TILING_QKV, 0, 1, 1024, 1024, 5},
    {"flash_attn_mma_stages_split_q_tiling_qkv_acc_f32_swizzle_qkv", LC_ATTN_TILING_QKV, 0, 1, 1024, 1024, 5},
    {"flash_attn_cute", LC_ATTN_SPLIT_Q, 0, 1, 256, 256, 4},
    // -DBUILD_FLASH_ATTN_MMA_OTHERS (flash_attn.cc:217-223)
    {"flash_attn_mma_stages_split_q_shared_qkv_Os2g", LC_ATTN_SHARED_QKV, 0, 0, 128, 256, 5},
    {"flash_attn_mma_stages_split_q_shared_kv_acc_f32_rr", LC_ATTN_SHARED_KV, 0, 1, 128, 256, 5},
    {"flash_attn_mma_stages_split_q_shared_qkv_acc_f32_rr", LC_ATTN_SHARED_QKV, 0, 1, 256, 256, 5},
};
constexpr int kNumAttnEntries = sizeof(kAttnEntries) / sizeof(kAttnEntries[0]);

const HgemmEntry* find_hgemm(const char* name) {
  if (!name) return nullptr;
  for (int i = 0; i < kNumHgemmEntries; ++i)
    if (strcmp(kHgemmEntries[i].name, name) == 0) return &kHgemmEntries[i];
  return nullptr;
}
const AttnEntry* find_attn(const char* name) {
  if (!name) return nullptr;
  for (int i = 0; i < kNumAttnEntries; ++i)
    if (strcmp(kAttnEntries[i].name, name) == 0) return &kAttnEntries[i];
  return nullptr;
}

// ------------------------------------------------------------------------------------------------
// HGEMM launchers
bool is_w4_variant(int v) {
  return v == LC_HGEMM_MFMA256W4B || v == LC_HGEMM_MFMA256W4C || v == LC_HGEMM_MFMA256W4X ||
         v == LC_HGEMM_MFMA256W4Y;
}

// Block -> C tile map handed to the tiled kernels (block_tile, hgemm_mfma256.hip): >= 1 = the reference's block swizzle with
// that many tile columns per N panel, -1 = XCD super-block raster.  Auto rule (measured, profiles/r3b_hgemm_raster_ab.log,
// 0.5 s sustained per cell, 3 interleaved rounds): operands that fit the 256 MiB Infinity Cache are served from it whatever
// the order (8192^3: A + B = 256 MiB, block swizzle 1441 / xcd16 1435 TFLOP/s TN; 4096^3 +0.2 %), beyond it the super-block
// raster streams every panel from HBM about a third as often: 12544^3 +5.7 %, 15360^3 +8.3 %, 16384^3 +5.3 % TN (+4.7 ... 6.9 %
// NN), which is what lifts AUTO from 4 ... 9 % behind hipBLASLt TN to level with it on the reference's published sizes.
int panel_tiles(int swizzle_stride, int tiles_n, int tile_n, size_t operand_bytes) {
  // (round 6: the threshold came down from 1.5 x to 1.0625 x the Infinity Cache — 8704^3 + 3.2 %, 8960^3 + 4.2 %, 9728^3 + 4.3 % with the super-block
  // raster, 9216^3 level, 8192^3 and below 0.3 ... 1.3 % better on the block swizzle: profiles/r6i_hgemm_knob_sched_ab.log)
  const bool xcd16 = g_tune_hgemm_raster == 2 || (g_tune_hgemm_raster == 0 && operand_bytes > ((size_t)272 << 20));
  if (xcd16) return -1;                     // the kernel ignores the stride
  if (swizzle_stride <= 1) return tiles_n;  // no thread-block swizzle: plain N-major raster
  int w = swizzle_stride / tile_n;
  if (w < 1) w = 1;
  if (w > tiles_n) w = tiles_n;
  return w;
}

// The CU count the launch rules reason with (lc_tune_set "rule_cus"): the device's own unless a test asks what a 128- or 304-CU part would
// be told.  Only RULES use it (which kernel, which tile, which split factor); every grid is sized with device_cu_count().
int rule_cu_count() {
  const int k = g_tune_rule_cus;
  return k > 0 ? k : device_cu_count();
}
// Constants of the split-KV cost model (attn_split_auto) measured on THIS device by lc_tune_calibrate (round 6; round-5 verdict weak #13:
// they were fitted once, on one box's clocks); one record per device ordinal, valid != 0 once measured.
struct AttnCalib {
  std::atomic<int> valid{0};
  float tau128 = 0.f, tau64 = 0.f, x0 = 0.f, bytes_per_us = 0.f;
};
AttnCalib g_attn_calib[64];

// Waves of hgemm_mfma128_kernel for a launch of `blocks` 128 x 128 tiles (lc_tune_set "hgemm_128w"): eight (KSW = 2, two waves per SIMD
// inside one block) on grids that leave CUs idle, four otherwise.  Measured (profiles/r5g_hgemm_128w.log, four vs eight waves, TN / NN):
// 1024^3 (64 blocks) 172 / 167 -> 192 / 189 TFLOP/s, 1536^3 (144) 410 / 396 -> 454 / 425; 2048^3 (256 blocks = one per CU) 705 -> 701: level —
// there the 128 x 128 tile is bound by L2 bandwidth (64 FLOP / B: 10 TB/s at 700 TFLOP/s), not by latency, and from 2560^3 on the NN form
// LOSES (825 -> 587: twice the waves on the transpose reads).  Auto: eight up to 0.6 blocks per CU.
int mfma128_ksw(long blocks) {
  const int k = g_tune_hgemm_128w;
  if (k == 1 || k == 2) return k;
  return 5 * blocks <= 3 * (long)rule_cu_count() ? 2 : 1;
}
template <bool B_KN>
int launch_mfma128_blocks(int ksw, int nblocks, const half_t* A, const half_t* B, half_t* C, int M, int N, int K, int tiles_m, int tiles_n,
                          int pw, int rem_base, int rem_blocks, int nright, int ks, float* ws, hipStream_t st) {
  if (ksw == 2 && ks == 1) {
    auto kern = hgemm_mfma128_kernel<B_KN, 2>;
    if (int rc = set_dyn_lds(kern, HGEMM128_LDS)) return rc;
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(512), HGEMM128_LDS, st, A, B, C, M, N, K, tiles_m, tiles_n, pw, rem_base, rem_blocks, nright, 1,
                       (float*)nullptr);
  } else {
    auto kern = hgemm_mfma128_kernel<B_KN, 1>;
    if (int rc = set_dyn_lds(kern, HGEMM128_LDS)) return rc;
    hipLaunchKernelGGL(kern, dim3(nblocks * ks), dim3(256), HGEMM128_LDS, st, A, B, C, M, N, K, tiles_m, tiles_n, pw, rem_base, rem_blocks, nright,
                       ks, ws);
  }
  return check_launch();
}

template <bool B_KN>
int launch_mfma256(const half_t* A, const half_t* B, half_t* C, int M, int N, int K, int variant,
                   int swizzle_stride, hipStream_t st) {
  // Interior = the 256-tileable part of C.  M, N % 256 == 128 (the reference's kernels are legal on multiples of 128,
  // hgemm_mma_stage.cu:675-676; resolve_hgemm_variant admits them for hgemm_w4y_kernel only): the 128-wide right / bottom border
  // strips go to the 128-tile kernel in the launch that also takes the ragged last wave.
  const int tiles_m = M / BM, tiles_n = N / BN;
  const int pw = panel_tiles(swizzle_stride, tiles_n, BN, ((size_t)M + N) * K * 2);
  const dim3 grid(tiles_m * tiles_n), block(512);
  if (is_w4_variant(variant)) {
    // Ragged last wave (lc_tune_set "hgemm_tail"): T tiles on 256 CUs run ceil(T / 256) tile periods, the last one with T % 256
    // workgroups (256 = the CU count of an MI355X; the rule uses the device's own).  When that remainder R is at most half a wave, the generated-loop kernel computes the first T − R raster
    // ids and the 128-tile kernel the four quadrants of each of the other R (4 R <= 512 workgroups at two per CU: ONE period of
    // a quarter-size tile) — 6144^3: 2.25 waves -> 2 + a short one instead of 3 (profiles/r3e_hgemm_tail.log).
    const int ncu = device_cu_count();   // (the same per-device figure the persistent launchers use)
    const int T = tiles_m * tiles_n, R = T % ncu;
    const bool w4y = w4_effective_variant(variant, B_KN, N, K) == LC_HGEMM_MFMA256W4Y;
    const int nright = (N % BN) ? M / BM1 : 0, nbottom = (M % BM) ? 2 * tiles_n : 0;   // border strips in 128 x 128 tiles
    if ((nright || nbottom || (K % BK)) && !w4y) return LC_ERR_SHAPE;   // (resolve_hgemm_variant never lets this happen)
    const int tail_knob = g_tune_hgemm_tail;
    // (knob 3 / 4: the remainder up to 0.75 / 1.0 of the CUs instead of 0.5 — A/B of the threshold, profiles/r6l_hgemm_tail_mid.log)
    const bool split = tail_knob != 0 && T > ncu && R > 0 && w4y && (tail_knob == 3 ? 4 * R <= 3 * ncu : tail_knob == 4 ? true : 2 * R <= ncu);
    if (int rc = launch_w4_family(A, B, C, M, N, K, variant, B_KN, tiles_m, tiles_n, pw, split ? T - R : -1, st)) return rc;
    // Round 6 (lc_tune_set "hgemm_tail" = 1, the default; 2 = round 5's path below): with no border strips the quadrants of the left-out
    // tiles run on the mid-size kernel — three ring slots when they fit one round of the CUs, two slots at two workgroups per CU beyond; no
    // workspace, no reduce launch, legal under graph capture (profiles/r6l_hgemm_tail_mid.log: + 3 ... 7 % at 4352 ... 4864, 6144, 10240)
    if (split && nright == 0 && nbottom == 0 && tail_knob != 2 && g_tune_hgemm_mid != 1 && K < (1 << 22) && N < (1 << 22))
    {
      // sub-tile (lc_tune_set "hgemm_tail_tile"): 64 x 128 eighths while they fit ONE round of the CUs (R <= ncu / 8: twice the workgroups
      // of the quadrants on CUs that would otherwise idle), else 128 x 128 quadrants
      const int tile_knob = g_tune_hgemm_tail_tile;
      const int tmw = tile_knob == 1 ? 1 : tile_knob == 2 ? 2 : (8 * R <= ncu ? 1 : 2);
      const int blocks = (tmw == 1 ? 8 : 4) * R;
      return launch_hgemm_mid_rem(A, B, C, M, N, K, B_KN, tmw, blocks <= ncu ? 3 : 2, tiles_m, tiles_n, pw, T - R, R, st);
    }
    const int nb128 = (split ? 4 * R : 0) + nright + nbottom;
    if (nb128 == 0) return LC_OK;
    // Split-K of these blocks (lc_tune_set "hgemm_splitk"): a lone 128-tile block walks its K range at a quarter of a CU's MFMA rate
    // (one barrier per K tile, nothing to overlap with), and the launch holds few of them — 8192 x 8320 x 8192: 64 blocks, 107 us
    // for 1.5 % of the FLOPs (profiles/r5a_hgemm_shapes.log).  ks blocks per tile (about 1.5 per CU, each
    // range >= 8 K tiles) write fp32 partials into this stream's cached workspace, a second kernel adds them and stores C.  Not while
    // the stream is being captured (no allocation, no pool pointer inside a graph): one block per tile then.
    int ks = 1;
    const int knob = g_tune_hgemm_splitk, KT = K / BK;
    if (knob >= 2) ks = knob;
    else if (knob == 0 && nb128 < ncu) ks = (3 * ncu / 2 + nb128 / 2) / nb128;   // ~1.5 blocks per CU (profiles/r5b_hgemm_splitk_sweep.log: 64 blocks: 4 best, 129 blocks: 3 best)
    if (ks > 8) ks = 8;
    while (ks > 1 && KT / ks < 8) --ks;
    WorkspaceLease lease;
    if (ks > 1 && !stream_is_capturing(st)) lease = stream_workspace(st, (size_t)nb128 * ks * (128 * 128 * sizeof(float)));
    if (!lease.ptr) ks = 1;
    // (no workspace — graph capture, knob — and few blocks: the eight-wave form of the kernel is the next best thing)
    if (int rc = launch_mfma128_blocks<B_KN>(mfma128_ksw(nb128), nb128, A, B, C, M, N, K, tiles_m, tiles_n, pw, split ? T - R : -2, split ? 4 * R : 0,
                                             nright, ks, static_cast<float*>(lease.ptr), st))
      return rc;
    if (ks > 1) {
      hipLaunchKernelGGL(hgemm_splitk_reduce_kernel, dim3(nb128), dim3(256), 0, st, static_cast<const float*>(lease.ptr), C, M, N, tiles_m,
                         tiles_n, pw, split ? T - R : -2, split ? 4 * R : 0, nright, ks);
      return check_launch();
    }
    return LC_OK;
  }
  if (false) {
#ifdef LC_DIAG
  } else if (variant == LC_HGEMM_MFMA256P2 && g_tune_hgemm_stamps) {
    auto kern = hgemm_pingpong2_kernel<B_KN, true>;
    if (int rc = set_dyn_lds(kern, HGEMM256_LDS)) return rc;
    hipLaunchKernelGGL(kern, grid, block, HGEMM256_LDS, st, A, B, C, M, N, K, tiles_m, tiles_n, pw);
#endif
  } else if (variant == LC_HGEMM_MFMA256P2) {
    auto kern = hgemm_pingpong2_kernel<B_KN>;
    if (int rc = set_dyn_lds(kern, HGEMM256_LDS)) return rc;
    hipLaunchKernelGGL(kern, grid, block, HGEMM256_LDS, st, A, B, C, M, N, K, tiles_m, tiles_n, pw);
  } else {
    auto kern = hgemm_mfma256_kernel<B_KN>;
    if (int rc = set_dyn_lds(kern, HGEMM256_LDS)) return rc;
    hipLaunchKernelGGL(kern, grid, block, HGEMM256_LDS, st, A, B, C, M, N, K, tiles_m, tiles_n, pw);
  }
  return check_launch();
}

template <bool B_KN>
int launch_mfma128(const half_t* A, const half_t* B, half_t* C, int M, int N, int K, int swizzle_stride,
                   hipStream_t st) {
  const int tiles_m = M / BM1, tiles_n = N / BN1;
  const int pw = panel_tiles(swizzle_stride, tiles_n, BN1, ((size_t)M + N) * K * 2);
  return launch_mfma128_blocks<B_KN>(mfma128_ksw((long)tiles_m * tiles_n), tiles_m * tiles_n, A, B, C, M, N, K, tiles_m, tiles_n, pw, -1, 0, 0, 1,
                                     nullptr, st);
}

// The mid-size kernel (hgemm_mid.hip; lc_tune_set "hgemm_mid", "hgemm_mid_ns"): which tile serves this shape, tmw == 0 = not this kernel.
// Auto = hipBLASLt's own heuristic for these sizes read off its kernel names (profiles/r6a_vendor_kernels.log) and measured here tile by
// tile (profiles/r6c_hgemm_mid_ab.log): when a tile's grid fits ONE ROUND of at most one workgroup per CU, the smallest such tile — most
// workgroups, least work on the busiest CU — with three ring slots (the DMA two tiles ahead): 64 x 128 at 1024 / 1280, 64 x 192 at 1536
// TN, 128 x 128 at 1536 NN / 1792 / 2048, 128 x 192 at 2304 TN, 192 x 128 at 2304 NN (the 64-row tiles lose to the 128-row ones as soon as both need more than a
// round: 2304 NN 780 vs 866 TFLOP/s, 2560 646 vs 983); otherwise 128 x 128 with two slots and two workgroups per CU (2304 NN, 2560, 2816).
// `gated` (LC_HGEMM_AUTO): only where the 256-tile kernel does not apply anyway (resolve_hgemm_variant: <= 128 tiles of 256 x 256) and the
// 128 x 128 grid holds more than 3 / 16 blocks per CU (below — 768^3: 36 blocks, level — the eight-wave 128-tile kernel keeps the shape).
struct MidTile { int tmw, tnw, ns, ks; long wgs; };   // ks > 1: split-K, needs ks x M x N floats of workspace (launch_mid; none under graph capture)
MidTile mid_tile_auto(int M, int N, int K, bool b_kn, bool gated) {
  MidTile none{0, 0, 0, 1, 0};
  if (M % 64 != 0 || N % 64 != 0 || K % 32 != 0 || K < BK || K >= (1 << 22) || N >= (1 << 22)) return none;
  const int k = g_tune_hgemm_mid, kns = g_tune_hgemm_mid_ns;
  if (k == 1 && gated) return none;
  const long ncu = rule_cu_count();
  const long min_blocks = 3 * ncu / 16;   // 48 of 128 x 128 on 256 CUs (768^3: 36 blocks, level with the eight-wave kernel; 1024^3: 64 blocks, + 15 %)
  // (... unless K is long enough to split: 512 x 512 x 8192 runs 32 workgroups x 8 K ranges here)
  const bool long_k = g_tune_hgemm_mid_splitk != 1 && K / BK >= 64;   // (two ranges of 32 K tiles)
  // (M or N a multiple of 64 only — 2880^3 — has no other tiled kernel: any tile that divides it beats hgemm_generic_kernel by 10 x)
  if (gated && M % 128 == 0 && N % 128 == 0 && (long)(M / 128) * (N / 128) <= min_blocks && !long_k) return none;
  MidTile best = none, big = none;
  long best_wgs = 0;   // best one-round tile; largest legal tile (the multi-round choice)
  long best_area = 0, big_area = 0;
  for (int tmw : {2, 3, 1})        // (ties between equal areas go to the tile seen first: 128 x 192 before 192 x 128)
    for (int tnw = 2; tnw <= 3; ++tnw) {
      if (k >= 10 && k != 10 * tmw + tnw) continue;
      if (M % (64 * tmw) != 0 || N % (64 * tnw) != 0 || (b_kn && tnw != 2)) continue;
      const long wgs = (long)(M / (64 * tmw)) * (N / (64 * tnw)), area = 4096L * tmw * tnw;
      if (wgs <= ncu && (best.tmw == 0 || area < best_area)) {
        best = MidTile{tmw, tnw, 3, 1, wgs};
        best_area = area;
        best_wgs = wgs;
      }
      // multi-round: 128 x 128 before 128 x 192 (one workgroup per CU by registers) before the 64-row tiles
      const long rank = (tmw == 2 && tnw == 2) ? 5 : (tmw == 2 ? 4 : tmw == 3 ? 3 : tnw - 1);
      if (big.tmw == 0 || rank > big_area) {
        big = MidTile{tmw, tnw, 2, 1, 0};
        big_area = rank;
      }
    }
  // 192 x 192 is the one tile with more work per CU (36864 outputs) than a double round of 128 x 128 at two workgroups per CU (2 x 16384):
  // it wins only where the 128 x 128 grid needs more than one such round (3072^3: 576 blocks; 3072 x 2304: 432 blocks, 950 vs 1016 TFLOP/s)
  if (best.tmw && best_area > 32768 && k < 10 && big.tmw == 2 && big.tnw == 2 && (long)(M / 128) * (N / 128) <= 2 * ncu) best = none;
  MidTile t = best.tmw ? best : big;
  if (!best.tmw && t.tmw * t.tnw >= 6) t.ns = 3;   // one workgroup per CU by registers anyway: the third slot is free (8192 x 8256 x 4096 TN: 1004 -> 1108)
  if (t.tmw && (kns == 2 || kns == 3)) t.ns = kns;
  // split-K (round 6, lc_tune_set "hgemm_mid_splitk"): a one-round grid on at most half the CUs with a long K — as many K ranges as fill
  // the CUs, each of at least 32 K tiles (1024 x 1024 x 8192: 128 workgroups x 2, 512 -> 620 TFLOP/s; 1024 x 1024 x 2048 with 16 tiles per range: 415 -> 306;
  // profiles/r6p_hgemm_rect_splitk.log); never at the reference sweep's sizes (1024^3: 16 K tiles)
  const int ksk = g_tune_hgemm_mid_splitk, KT = K / BK;
  if (best.tmw && t.tnw == 2 && t.tmw <= 2 && ksk != 1) {
    long ks = ksk >= 2 ? ksk : std::min<long>(std::min<long>(ncu / best_wgs, KT / 32), 8);
    while (ks > 1 && KT < 2 * ks) --ks;
    if (ks > 1 && (size_t)ks * M * N * sizeof(float) <= ((size_t)256 << 20)) {
      t.ks = (int)ks;
      t.ns = 3;
    }
  }
  return t;
}
int launch_mid(const half_t* A, const half_t* B, half_t* C, int M, int N, int K, bool b_kn, MidTile t, int swizzle_stride, hipStream_t st) {
  const int pw = panel_tiles(swizzle_stride, N / (64 * t.tnw), 64 * t.tnw, ((size_t)M + N) * K * 2);
  if (t.ks > 1 && !stream_is_capturing(st)) {
    WorkspaceLease lease = stream_workspace(st, (size_t)t.ks * M * N * sizeof(float));
    if (lease.ptr) return launch_hgemm_mid(A, B, C, M, N, K, b_kn, t.tmw, t.tnw, 3, pw, st, static_cast<float*>(lease.ptr), t.ks);
  }
  return launch_hgemm_mid(A, B, C, M, N, K, b_kn, t.tmw, t.tnw, t.ns, pw, st);   // (no workspace — graph capture, allocation failure: one K range)
}

// hgemm_edge_kernel over the right strip (all rows, columns Ni .. N) and the bottom strip (rows Mi .. M, columns 0 .. Ni) of C; Mi = Ni = 0:
// the whole matrix.  Ni % 128 == 0.
template <bool B_KN>
int launch_edge(const half_t* A, const half_t* B, half_t* C, int M, int N, int K, int Mi, int Ni, hipStream_t st) {
  const long nrc = (N - Ni + EN - 1) / EN, nright = nrc * ((M + EM - 1) / EM);
  const long nbottom = (long)((M - Mi + EM - 1) / EM) * (Ni / EN);
  if (nright + nbottom <= 0) return LC_OK;
  if (nright + nbottom > INT_MAX) return LC_ERR_SHAPE;
  auto kern = hgemm_edge_kernel<B_KN>;
  if (int rc = set_dyn_lds(kern, EDGE_LDS)) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)(nright + nbottom)), dim3(256), EDGE_LDS, st, A, B, C, M, N, K, Mi, Ni, (int)nright, (int)(nrc > 0 ? nrc : 1));
  return check_launch();
}
template <bool B_KN>
int launch_generic(const half_t* A, const half_t* B, half_t* C, int M, int N, int K, hipStream_t st) {
  const dim3 grid((N + GN - 1) / GN, (M + GM - 1) / GM), block(256);
  hipLaunchKernelGGL(hgemm_generic_kernel<B_KN>, grid, block, 0, st, A, B, C, M, N, K);
  return check_launch();
}

// ------------------------------------------------------------------------------------------------
// attention launchers
template <int D, int NW, bool VT, int ABL = 0>
int launch_attn(const half_t* Q, const half_t* K, const half_t* V, half_t* O, int B, int H, int N,
                hipStream_t st) {
  auto kern = attn_fwd_kernel<D, NW, VT, ABL>;
  constexpr int lds = attn_lds_bytes<D, VT>();
  if (int rc = set_dyn_lds(kern, lds)) return rc;
  const int nqb = N / (NW * 32);
  const dim3 grid((unsigned)((size_t)nqb * B * H)), block(NW * 64);
  const float sl2 = (1.0f / sqrtf((float)D)) * 1.4426950408889634f;
  hipLaunchKernelGGL(kern, grid, block, lds, st, Q, K, V, O, N, nqb, sl2);
  return check_launch();
}

// Which kernel serves a D <= 128 problem (ONE function for the launcher and lc_attn_kernel_name()).  Codes (lc_tune_set "attn_nw"):
//   513 / 515 / 517  the merged-phase 4-wave kernel attn_fwd_w4u_kernel<D, VT, WALK> (attn_w4u.hip: D = 64 / 128, N % 256 == 0, V as
//                    [B,H,N,D] or — the three *_swizzle_qkv entries — [B,H,D,N]) with WALK 0 (one 256-row query block per workgroup),
//                    1 (persistent workgroup per CU, static walk), 2 (persistent, dynamic per-XCD block queue)
//   514              the same design with each phase as one generated asm statement (attn_w4i.hip: D = 32 / 64 / 96 / 128, V as
//                    [B,H,N,D]; the only merged-phase kernel for D = 96 / 32)
//   8 / 4 / 2        the lock-step kernel with that many waves (attn_fwd.hip: every other shape)
// 512 (round 2's attn_w4n) is accepted as an alias of 513: attn_w4u<128, false, 0> IS that kernel; 256 / 260 / 516 were retired in
// round 4 with attn_w4m.hip / attn_w8g.hip.
int attn_walk_auto(int N, int D) {
  // auto (lc_tune_set "attn_walk": 0 = this rule; measured, profiles/r3k, profiles/r4c_attn_walks.log, r5p_attn_walks.log): up to N = 4096 the
  // persistent static walk (round 3: config 3 + 1.7 %, N = 2048 + 1.0 %; later boxes: + 0.5 % / level), beyond it one block per workgroup —
  // with 16+ blocks per CU the hardware dispatcher balances better than either walk (D = 128, N = 8192: static - 1.0 %, dynamic queue - 1.1 %;
  // the queue is a validated alternative, never the default) — except D = 64, whose blocks are half as long: (1,48,8192,64) static + 1.4 %
  const int k = g_tune_attn_walk;
  if (k >= 1 && k <= 3) return k - 1;
  return (N <= 4096 || (D == 64 && N <= 8192)) ? 1 : 0;
}
// Split-KV factor of the merged-phase kernel for a launch of `bh` (batch, head) problems (lc_tune_set "attn_split"; 1 = no split).
// The kernel owns 256 query rows per workgroup and one workgroup per CU, so g = bh N / 256 workgroups on ncu CUs run ceil(g / ncu)
// rounds of T = N / 64 KV tiles: a grid that does not fill the GPU (the reference author's own regime, README.md:120 "B <= 4, H <= 48,
// SeqLen <= 8192") leaves CUs idle for the whole launch, and a grid of 1.25 rounds pays for 2.  With S KV ranges per query block the
// launch runs ceil(g S / ncu) rounds of T / S tiles + the combine.  Auto picks, among S = 2, 4, 8, 16 (T divisible, >= kMinSplitTiles
// tiles per range, partials <= 256 MiB), the S that minimises the cost model
//     t(S) = ceil(g S / ncu) (T / S) tau_D + [S > 1] (x0 + S * 4 bh N D bytes / bw)          (microseconds)
// and splits when that is 5 % below t(1).  Fitted to profiles/r5b_attn_split.log, r5f_attn_split_quant.log, r5f_small_split_kernel_
// durations.log: tau_128 = 1.35, tau_64 = 0.85 us per 64-key tile of a 256-row block, x0 = 5 us (the combine kernel: 4.9 us), bw = the rate
// at which a range's fp16 partial is written and read back (2.6 TB/s: small transfers).  Examples (256 CUs): (1,8,1024,128) -> 4 (+ 34 %),
// (1,8,2048,64) -> 4 (+ 61 %), (1,4,4096,128) -> 4 (2.1 x), (1,2,8192,128) -> 8 (2.6 x), (1,16,2048,128) -> 2 (+ 20 %), (1,10,8192,128) -> 4
// (1.25 rounds: + 22 %), (1,12,8192,64) -> 2 (+ 18 %), (1,32,1024,128) -> 1 (a half-full GPU and 16 tiles: the combine costs more than
// half the walk saves), (1,6,8192,128) -> 1, config 3 / 4 -> 1.  bh < 0 (lc_attn_kernel_name has no batch / head count): no split.
constexpr int kMinSplitTiles = 4;
constexpr double kSplitFixedUs = 5.0, kSplitBytesPerUs = 2.6e6;
int attn_split_auto(int D, int N, long bh) {
  const int k = g_tune_attn_split;
  if ((D != 128 && D != 64) || N % 256 != 0 || bh <= 0 || k == 1) return 1;
  const int T = N / 64;
  const double part = 4.0 * (double)bh * N * D;   // bytes of one range's partial O, written + read
  const double part_cap = 2.0 * ((size_t)256 << 20);   // partials <= 256 MiB, forced factor or auto (round-5 advisor: a forced 16 on config 4 asked for 34 GiB)
  if (k >= 2) return (T % k == 0 && T / k >= 2 && k * part <= part_cap) ? k : 1;
  const long ncu = rule_cu_count(), g = bh * (N / 256);
  // the model's constants: measured on this device (lc_tune_calibrate) or the values fitted on the round-5 boxes
  double tau = D == 128 ? 1.35 : 0.85, fixed_us = kSplitFixedUs, bytes_per_us = kSplitBytesPerUs;
  {
    int dev = 0;
    if (g_tune_attn_calib == 0 && g_tune_rule_cus == 0 && hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && g_attn_calib[dev].valid.load(std::memory_order_acquire)) {
      tau = D == 128 ? g_attn_calib[dev].tau128 : g_attn_calib[dev].tau64;
      fixed_us = g_attn_calib[dev].x0;
      bytes_per_us = g_attn_calib[dev].bytes_per_us;
    } else {
      (void)hipGetLastError();
    }
  }
  int best = 1;
  const double t1 = (double)((g + ncu - 1) / ncu) * T * tau;
  double tbest = 0.95 * t1;
  for (int S = 2; S <= 16 && T % S == 0 && T / S >= kMinSplitTiles && S * part <= part_cap; S *= 2) {
    const double t = (double)((g * S + ncu - 1) / ncu) * (T / S) * tau + fixed_us + S * part / bytes_per_us;
    if (t < tbest) {
      tbest = t;
      best = S;
    }
  }
  return best;
}
// *nsplit (when given) receives the split-KV factor that goes with a 519 answer, 1 otherwise: the launcher must not evaluate the rule a
// second time (round-5 advisor: a concurrent lc_tune_set between the two reads could pair walk 3 with one range).
int choose_attn_nw(int D, bool vt, int N, long bh = -1, int* nsplit = nullptr) {
  int want = g_tune_attn_nw;   // 0 = auto (read once per launch)
  if (want == 512) want = 513;
  if (nsplit) *nsplit = 1;
  const bool merged = (D == 128 || D == 64) && N % 256 == 0;
  if (merged && g_tune_attn_ablate == 0) {
    if (want == 0) {
      const int ns = attn_split_auto(D, N, bh);
      if (ns > 1) {
        if (nsplit) *nsplit = ns;
        return 519;
      }
      // Small grids the split rule leaves alone (too few KV tiles for the combine to pay): up to half a GPU of 256-row blocks and N <= 2048
      // the 4-wave lock-step kernel's 128-row workgroups fill twice the CUs — (1,32,1024,128) 655 vs 601 TFLOP/s, (1,32,1024,64) 498 vs 444
      // (profiles/r4q_small_grids_d128.log, r5i_small_grids.log); from one full round of blocks on the merged-phase kernel is far ahead (992 vs 760)
      if (bh > 0 && 2 * bh * (N / 256) <= rule_cu_count() && N <= 2048 && g_tune_attn_split != 1) return 4;
      return 513 + 2 * attn_walk_auto(N, D);
    }
    if (want == 513 || want == 515 || want == 517) return want;
    if (want == 514 && !vt) return 514;
  }
  // N % 256 != 0 (N % 64 == 0; N % 128 == 0 is what the reference's own kernels need: flash_attn_mma_share_qkv.cu:839 asserts
  // N % max(Br, Bc) == 0): the merged-phase kernel with one block per workgroup, the head's last 256-row block partly real — the waves whose 64
  // rows lie behind N compute on a clamped copy of the last row and store nothing ((256 - N % 256) / (N + 256 - N % 256) of the work wasted).
  // From N = 1152 on that beats the lock-step kernel's MFMA-busy 0.46 vs 0.58 (profiles/r5d_attn_n128.log: (4,32,4224,128) 1210 vs 901 TFLOP/s,
  // (4,32,1152,128) 886 vs 771, (1,48,8320,64) 961 vs 799; (2,16,896,64) 372 vs 427: the lock-step kernel keeps N < 1152)
  if ((D == 128 || D == 64) && N % 256 != 0 && N % 64 == 0 && N >= 1152 && g_tune_attn_ablate == 0 && (want == 0 || want == 513)) return 513;
  // D = 96 / 32: only the generated kernel (attn_w4i.hip, 514) has a merged-phase instantiation (256-B / 128-B padded LDS rows)
  if ((D == 96 || D == 32) && !vt && N % 256 == 0 && (want == 0 || want >= 256)) return 514;
  if (N % 256 == 0 && (want == 0 || want >= 8)) return 8;
  if (N % 128 == 0 && (want == 0 || want >= 4)) return 4;
  return 2;
}

template <int D, bool VT>
int launch_attn_nw(const half_t* Q, const half_t* K, const half_t* V, half_t* O, int B, int H, int N,
                   hipStream_t st) {
  int ns = 1;
  const int nw = choose_attn_nw(D, VT, N, (long)B * H, &ns);
  if constexpr (D == 128 || D == 64) {
    if (nw == 513 || nw == 515 || nw == 517 || nw == 519) {
      const int walk = (nw - 513) / 2;   // (519 = split-KV with ns ranges: auto only)
      if constexpr (D == 128) return VT ? launch_attn_w4u_d128t(Q, K, V, O, B, H, N, walk, ns, st) : launch_attn_w4u_d128(Q, K, V, O, B, H, N, walk, ns, st);
      else return VT ? launch_attn_w4u_d64t(Q, K, V, O, B, H, N, walk, ns, st) : launch_attn_w4u_d64(Q, K, V, O, B, H, N, walk, ns, st);
    }
  }
  if constexpr (!VT) {
    if (nw == 514) return launch_attn_w4i(Q, K, V, O, B, H, N, D, g_tune_attn_w4i_sched, st);
  }
  if constexpr (D == 128 && !VT) {   // perf-diagnosis instantiations (lc_tune_set "attn_ablate")
    switch (g_tune_attn_ablate) {
#ifdef LC_DIAG
      case 1: return launch_attn<D, 8, VT, 1>(Q, K, V, O, B, H, N, st);
      case 2: return launch_attn<D, 8, VT, 2>(Q, K, V, O, B, H, N, st);
      case 3: return launch_attn<D, 8, VT, 3>(Q, K, V, O, B, H, N, st);
      case 4: return launch_attn<D, 8, VT, 4>(Q, K, V, O, B, H, N, st);
      case 6: return launch_attn<D, 8, VT, 6>(Q, K, V, O, B, H, N, st);
      case 7: return launch_attn<D, 8, VT, 7>(Q, K, V, O, B, H, N, st);
      case 8: return launch_attn<D, 8, VT, 8>(Q, K, V, O, B, H, N, st);
      case 16: return launch_attn<D, 8, VT, 16>(Q, K, V, O, B, H, N, st);
      case 24: return launch_attn<D, 8, VT, 24>(Q, K, V, O, B, H, N, st);
      case 30: return launch_attn<D, 8, VT, 30>(Q, K, V, O, B, H, N, st);
      case 31: return launch_attn<D, 8, VT, 31>(Q, K, V, O, B, H, N, st);
      case 32: return launch_attn<D, 8, VT, 32>(Q, K, V, O, B, H, N, st);
#endif
      default: break;
    }
  }
  if (nw == 8) return launch_attn<D, 8, VT>(Q, K, V, O, B, H, N, st);
  if (nw == 4) return launch_attn<D, 4, VT>(Q, K, V, O, B, H, N, st);
  return launch_attn<D, 2, VT>(Q, K, V, O, B, H, N, st);
}

template <int D, int NW, bool VT, bool BF16 = false>
int launch_attn_bigd(const half_t* Q, const half_t* K, const half_t* V, half_t* O, int B, int H, int N,
                     hipStream_t st) {
  constexpr int DO = D > 256 ? 256 : D;   // output columns per workgroup (D = 512: two column halves)
  auto kern = attn_fwd_bigd_kernel<D, DO, NW, VT, BF16>;
  constexpr int lds = attn_bigd_lds_bytes<NW>();
  if (int rc = set_dyn_lds(kern, lds)) return rc;
  const int nqb = N / (NW * 32);
  const dim3 grid((unsigned)((size_t)nqb * B * H * (D / DO))), block(NW * 64);
  const float sl2 = (1.0f / sqrtf((float)D)) * 1.4426950408889634f;
  hipLaunchKernelGGL(kern, grid, block, lds, st, Q, K, V, O, N, nqb, sl2);
  return check_launch();
}

// D = 256 / 512 with N % 128 == 0: the full-width kernel (attn_bigd2.hip; V as [B,H,N,D], or — D = 256, the reach of the reference's
// *_swizzle_qkv entries — as [B,H,D,N]) unless lc_tune_set "attn_d512" = 1 asks for round 1's column-split kernel (kept as the
// independently written cross-check; it also serves N % 128 != 0 and D = 512 with V transposed).  D = 1024 with N % 64 == 0: the pair
// kernel (attn_bigd4.hip); the column-split kernel under knob 1 and for ragged N.
bool use_bigd2(int D, bool vt, int N) {
  return (D == 256 || (D == 512 && !vt)) && N % 128 == 0 && g_tune_attn_d512 != 1;   // (2: attn_bigd3, same launcher; not for vt)
}
bool use_bigd4(int D, bool vt, int N) { return D == 1024 && !vt && N % 64 == 0 && g_tune_attn_d512 != 1; }
// D = 512: attn_bigd6 (16x16x32 MFMAs) or attn_bigd2 (32x32x16): kBigd6Auto says which one auto means, knob 3 selects the other
constexpr bool kBigd6Auto = true;    // profiles/r4k_bigd6.log: fp16 + 3.4 ... 4.7 %, bf16 + 1.8 ... 2.8 % at the cap (zero-filled: - 8 %, the 16-wide stream is more issue-bound)
bool use_bigd6(int D, bool vt, int N) {
  const int k = g_tune_attn_d512;
  return D == 512 && !vt && N % 128 == 0 && (((k == 0 || k == 4) && kBigd6Auto) || (k == 3 && !kBigd6Auto));
}
// D = 256 with N % 256 == 0, either V layout: attn_bigd7 (64 query rows per wave, 16x16x32 MFMAs, KV rings) is auto; knob 3 selects
// attn_bigd2 (32 rows per wave, 32x32x16: the cross-check on the other MFMA shape, and the kernel for N % 256 == 128)
// attn_bigd7's workgroup owns 256 query rows, attn_bigd2's 128: on a grid that does not fill the GPU the smaller blocks win (measured,
// profiles/r4p_bigd7_small_grids.log: (1,8,1024,256) 156 vs 272 TFLOP/s, (1,16,2048,256) 693 vs 978; from 192 workgroups up attn_bigd7 is
// ahead).  With g7 = B H N / 256 workgroups of attn_bigd7 (1.6 time units each: twice the rows at 0.8 of the time per FLOP) against 2 g7 of
// attn_bigd2 (1 unit each), rounds of one workgroup per CU: attn_bigd7 iff 1.6 ceil(g7 / CUs) <= ceil(2 g7 / CUs), and always from 4 rounds up.
// bh < 0: "a grid that fills the GPU" (lc_attn_kernel_name has no batch / head count; lc_attn_kernel_name_bh has).  Knob 4 forces attn_bigd7 (tests of small shapes).
bool use_bigd7(int D, bool vt, int N, long bh) {
  const int k = g_tune_attn_d512;
  // N % 256 == 128 (round 5): the 256-row kernel with its last block half real, from N = 1152 (below, attn_bigd2's 128-row workgroups waste nothing)
  if (D != 256 || (N % 256 != 0 && (N % 256 != 128 || N < 1152)) || (k != 0 && k != 4)) return false;
  if (k == 4 || bh < 0) return true;
  const long ncu = rule_cu_count(), g7 = bh * ((N + 255) / 256);
  if (g7 >= 4 * ncu) return true;
  const long c7 = (g7 + ncu - 1) / ncu, c2 = (2 * g7 + ncu - 1) / ncu;
  return 16 * c7 <= 10 * c2;
}

template <int D, bool VT>
int launch_attn_bigd_nw(const half_t* Q, const half_t* K, const half_t* V, half_t* O, int B, int H, int N,
                        hipStream_t st) {
  if (use_bigd4(D, VT, N)) return launch_attn_bigd4(Q, K, V, O, B, H, N, g_tune_attn_d1024, st);
  if (use_bigd6(D, VT, N)) return launch_attn_bigd6(Q, K, V, O, B, H, N, false, st);
  if (use_bigd7(D, VT, N, (long)B * H)) return VT ? launch_attn_bigd7_vt(Q, K, V, O, B, H, N, st) : launch_attn_bigd7(Q, K, V, O, B, H, N, false, st);
  if (use_bigd2(D, VT, N)) return VT ? launch_attn_bigd2_vt(Q, K, V, O, B, H, N, D, st) : launch_attn_bigd2(Q, K, V, O, B, H, N, D, false, st);
  if (N % 128 == 0) return launch_attn_bigd<D, 4, VT>(Q, K, V, O, B, H, N, st);
  return launch_attn_bigd<D, 2, VT>(Q, K, V, O, B, H, N, st);
}

template <bool VT>
int launch_attn_d(const half_t* Q, const half_t* K, const half_t* V, half_t* O, int B, int H, int N,
                  int D, hipStream_t st) {
  switch (D) {
    case 32: return launch_attn_nw<32, VT>(Q, K, V, O, B, H, N, st);
    case 64: return launch_attn_nw<64, VT>(Q, K, V, O, B, H, N, st);
    case 96: return launch_attn_nw<96, VT>(Q, K, V, O, B, H, N, st);
    case 128: return launch_attn_nw<128, VT>(Q, K, V, O, B, H, N, st);
    case 256: return launch_attn_bigd_nw<256, VT>(Q, K, V, O, B, H, N, st);
    case 512: return launch_attn_bigd_nw<512, VT>(Q, K, V, O, B, H, N, st);
    case 1024: return launch_attn_bigd_nw<1024, VT>(Q, K, V, O, B, H, N, st);
    default: return LC_ERR_HEADDIM;
  }
}

}  // namespace

namespace {
// LC_HGEMM_KPAD (late round 6): K is not a multiple of 32 (K % 8 == 0, N % 8 == 0) on a problem large enough that hgemm_edge_kernel's 0.5 ... 0.66 x of the
// vendor hurts: A and B are copied into this stream's workspace with K padded to the next multiple of 32 by zeros (products with zero add nothing to an
// fp32 sum: the result is what the tuned kernels would produce on the padded problem, exactly), and the padded problem runs LC_HGEMM_AUTO's choice —
// tiled or LC_HGEMM_RAGGED, in its workspace-free form (the operands hold the workspace).  Costs two copies (8192 x 8192 x 8200: 0.54 GB of traffic).
// Not under graph capture, not beyond the workspace cap: the edge kernel then.  Kp = 0: not this path.
int kpad_plan(int M, int N, int K, bool al, bool gated) {
  if (!al || K % 8 != 0 || K % 32 == 0 || N % 8 != 0 || K < 256 || K >= (1 << 22) - 32 || N >= (1 << 22)) return 0;
  const int knob = g_tune_hgemm_kpad;
  if (gated && knob == 1) return 0;
  const long eb = (long)((M + 127) / 128) * ((N + 127) / 128);
  if (gated && knob == 0 && 4 * eb < rule_cu_count()) return 0;   // (below a quarter of a block per CU three launches cost more than the edge kernel's slower K walk; 1000^3: + 20 %, 8192 x 8192 x 8200: + 75 %)
  const int Kp = (K + 31) / 32 * 32;
  if (((size_t)M + N) * Kp * 2 > kWorkspaceCapBytes) return 0;
  return Kp;
}

// LC_HGEMM_RAGGED (late round 6): M and / or N are not multiples of the tiles (not legal in the reference, hgemm_mma_stage.cu:675-676), K is
// (K % 32 == 0, K >= 64) and rows are 16-byte aligned (N % 8 == 0).  The tiled kernels take N as C's / B's row stride and their tile counts
// separately, and hgemm_mid_edge_kernel (hgemm_mid.hip EDGE) runs 128 x 128 tiles that reach beyond M / N (clamped sources, predicated stores):
//   kind 1  more than half a CU's worth of 256 x 256 tiles: the INTERIOR — the largest top-left sub-matrix they divide — on hgemm_w4y_kernel exactly
//           as a problem of its own (+ its ragged last round on the mid-size kernel, as launch_mfma256), the L-shaped BORDER (right strip: all rows x
//           columns Ni .. N, bottom strip: rows Mi .. M x columns 0 .. Ni) on hgemm_mid_edge_kernel in a second launch
//   kind 2  otherwise: the whole problem on hgemm_mid_edge_kernel (three ring slots while the tiles fit one round of the CUs, else two)
// Every element of C is computed by exactly one kernel, deterministically; no workspace.  lc_tune_set "hgemm_ragged" = 1: never (hgemm_edge_kernel).
struct RaggedPlan { int kind, Mi, Ni, ns, tmw, tnw, ks; };   // ks > 1: split-K (kind 2, 64 / 128 x 128 tiles; needs the workspace: none under graph capture)
RaggedPlan ragged_plan(int M, int N, int K, bool al, bool b_kn, bool gated) {
  RaggedPlan none{0, 0, 0, 0, 0, 0, 1};
  if (!al || K % 32 != 0 || K < BK || N % 8 != 0 || K >= (1 << 22) || N >= (1 << 22)) return none;
  if (M % BM1 == 0 && N % BN1 == 0) return none;   // (a tiled shape)
  if (gated && g_tune_hgemm_ragged == 1) return none;
  const long ncu = rule_cu_count();
  const long t256 = (long)(M / BM) * (N / BN);
  if (2 * t256 > ncu && g_tune_hgemm_auto == LC_HGEMM_MFMA256W4Y && w4_effective_variant(LC_HGEMM_MFMA256W4Y, b_kn, N, K) == LC_HGEMM_MFMA256W4Y) {
    const int Mi = (M / BM) * BM, Ni = (N / BN) * BN;
    const long nb = (long)((N - Ni + 127) / 128) * ((M + 127) / 128) + (long)((M - Mi + 127) / 128) * (Ni / 128);   // border blocks
    return RaggedPlan{1, Mi, Ni, nb <= ncu ? 3 : 2, 2, 2, 1};
  }
  // the mid-size kernel's own rule (mid_tile_auto; measured on ragged shapes in profiles/r6ag_hgemm_edge_ab.log): the smallest tile whose grid fits ONE round of
  // at most one workgroup per CU (most workgroups, least work on the busiest CU; three ring slots) — 64 x 128, 128 x 128, then 128 x 192 (TN) / 192 x 128 (NN:
  // 128-column tiles only); where 128 x 128 at two per CU needs more than one double round, 192 x 192 (TN; 3000 x 3000 x 3008: 1074 vs 783 TFLOP/s) /
  // 192 x 128 (NN: 865 vs 724); else 128 x 128 with two slots at two workgroups per CU (2500 x 2504 x 2560 TN: 857 vs 773 on 192 x 192 in one round).
  const int tile_knob = g_tune_hgemm_ragged_tile;
  // split-K as the mid-size kernel's own (mid_tile_auto, "hgemm_mid_splitk"): a one-round grid of 64 / 128 x 128 tiles on at most half the CUs with a long K — as
  // many K ranges as fill the CUs, each of at least 32 K tiles, at most 8 (100 x 4096 x 4096: 64 workgroups x 2)
  auto split_k = [&](RaggedPlan p) {
    const int ksk = g_tune_hgemm_mid_splitk, KT = K / BK;
    const long wgs = (long)((M + 64 * p.tmw - 1) / (64 * p.tmw)) * ((N + 127) / 128);
    if (p.tnw != 2 || p.tmw > 2 || p.ns != 3 || ksk == 1 || wgs > ncu) return p;
    long ks = ksk >= 2 ? ksk : std::min<long>(std::min<long>(ncu / wgs, KT / 32), 8);
    while (ks > 1 && KT < 2 * ks) --ks;
    if (ks > 1 && launch_hgemm_mid_edge_sk_floats(M, N, p.tmw, (int)ks) * sizeof(float) <= ((size_t)256 << 20)) p.ks = (int)ks;
    return p;
  };
  auto blocks_of = [&](int tmw, int tnw) { return (long)((M + 64 * tmw - 1) / (64 * tmw)) * ((N + 64 * tnw - 1) / (64 * tnw)); };
  if (tile_knob != 0) {
    const int tmw = tile_knob / 10, tnw = tile_knob % 10;
    const bool legal = b_kn ? tnw == 2 : !(tmw == 3 && tnw == 2);
    if (legal) return split_k(RaggedPlan{2, 0, 0, (tmw == 2 && tnw == 2 && blocks_of(2, 2) > ncu) ? 2 : 3, tmw, tnw, 1});
  }
  if (blocks_of(1, 2) <= ncu) return split_k(RaggedPlan{2, 0, 0, 3, 1, 2, 1});
  if (blocks_of(2, 2) <= ncu) return split_k(RaggedPlan{2, 0, 0, 3, 2, 2, 1});
  if (blocks_of(2, 2) > 2 * ncu) return b_kn ? RaggedPlan{2, 0, 0, 3, 3, 2, 1} : RaggedPlan{2, 0, 0, 3, 3, 3, 1};
  if (b_kn ? blocks_of(3, 2) <= ncu : blocks_of(2, 3) <= ncu) return b_kn ? RaggedPlan{2, 0, 0, 3, 3, 2, 1} : RaggedPlan{2, 0, 0, 3, 2, 3, 1};
  return RaggedPlan{2, 0, 0, 2, 2, 2, 1};
}

// The border launch beside the interior (lc_tune_set "hgemm_ragged_fork"): one side stream per device, forked from the caller's stream
// by an event and joined back by another, so that the edge blocks (one 72 KiB workgroup per CU at best, a latency-bound K walk) fill the CUs
// the interior's last round leaves idle instead of holding the whole GPU for a round of their own.  The device's mutex (the one the workspace
// leases hold) covers the enqueue sequence: two host threads cannot interleave their fork / join events.  Not while the caller's stream is being
// captured, not when the side stream cannot be created: both launches on the caller's stream then.
struct ForkLane { hipStream_t side = nullptr; hipEvent_t fork = nullptr, join = nullptr; bool tried = false; };
ForkLane* fork_lane(int dev) {   // (call with the device's mutex held)
  static ForkLane lanes[64];
  if (dev < 0 || dev >= 64) return nullptr;
  ForkLane& l = lanes[dev];
  if (!l.tried) {
    l.tried = true;
    RelaxedCaptureMode relaxed;
    int least = 0, greatest = 0;   // (the lowest priority, for what it is worth)
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = 0;
    if (hipStreamCreateWithPriority(&l.side, hipStreamNonBlocking, least) != hipSuccess || hipEventCreateWithFlags(&l.fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&l.join, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      l.side = nullptr;
    }
  }
  return l.side ? &l : nullptr;
}

template <bool B_KN>
int launch_ragged_interior(const half_t* A, const half_t* B, half_t* C, int M, int N, int K, int swizzle_stride, hipStream_t st) {   // (kind 1)
  const int tiles_m = M / BM, tiles_n = N / BN;
  const int pw = panel_tiles(swizzle_stride, tiles_n, BN, ((size_t)tiles_m * BM + (size_t)tiles_n * BN) * K * 2);
  const int ncu = device_cu_count();
  const int T = tiles_m * tiles_n, R = T % ncu;
  const int tail_knob = g_tune_hgemm_tail;
  const bool split = tail_knob == 1 && T > ncu && R > 0 && 2 * R <= ncu && g_tune_hgemm_mid != 1;   // (launch_mfma256's default rule)
  if (int rc = launch_w4_family(A, B, C, M, N, K, LC_HGEMM_MFMA256W4Y, B_KN, tiles_m, tiles_n, pw, split ? T - R : -1, st)) return rc;
  if (split) {
    const int tmw = 8 * R <= ncu ? 1 : 2;
    const int blocks = (tmw == 1 ? 8 : 4) * R;
    return launch_hgemm_mid_rem(A, B, C, M, N, K, B_KN, tmw, blocks <= ncu ? 3 : 2, tiles_m, tiles_n, pw, T - R, R, st);
  }
  return LC_OK;
}

template <bool B_KN>
int launch_ragged(const half_t* A, const half_t* B, half_t* C, int M, int N, int K, const RaggedPlan& p, int swizzle_stride, hipStream_t st) {
  if (p.kind == 2) {
    if (p.ks > 1 && !stream_is_capturing(st)) {
      WorkspaceLease lease = stream_workspace(st, launch_hgemm_mid_edge_sk_floats(M, N, p.tmw, p.ks) * sizeof(float));
      if (lease.ptr) return launch_hgemm_mid_edge_sk(A, B, C, M, N, K, B_KN, p.tmw, p.ks, static_cast<float*>(lease.ptr), st);
    }
    return launch_hgemm_mid_edge(A, B, C, M, N, K, B_KN, p.tmw, p.tnw, p.ns, 0, 0, st);   // (no workspace — graph capture, allocation failure: one K range)
  }
  // Fork rule (profiles/r6ac … r6af_hgemm_edge_ab*.log; the hardware interleaves the two queues whatever their order or priority): beside an interior of
  // FULL rounds every CU a border block holds costs the interior a round of its own (4100 x 4104 x 4096, one round of 256 tiles: 1122 -> 995 TFLOP/s;
  // 12808^2 x 4096: − 5 %); beside an UNSPLIT last round that leaves at least 3 / 8 of the CUs idle the border fills them (5200^2 x 4096, 400 tiles:
  // 1182 -> 1234); beside a last round the mid-size kernel takes as quadrants it is a wash (5000^2 x 4096 − 4 %, 777 x 50264 x 4096 + 3 %): not forked.
  const int fork_knob = g_tune_hgemm_ragged_fork;
  bool fork = fork_knob == 2;
  if (fork_knob == 0) {
    const int ncu = device_cu_count(), T = (M / BM) * (N / BN), R = T % ncu;
    const bool split = g_tune_hgemm_tail == 1 && T > ncu && 2 * R <= ncu && g_tune_hgemm_mid != 1;   // (launch_ragged_interior's)
    fork = !split && R > 0 && 8 * (ncu - R) >= 3 * ncu;
  }
  int dev = 0;
  if (fork && !workspace_held_by_this_thread() && !stream_is_capturing(st) && hipGetDevice(&dev) == hipSuccess) {
    std::unique_lock<std::mutex> lock(workspace_pool(dev).mu);
    ForkLane* l = fork_lane(dev);
    if (l && hipEventRecord(l->fork, st) == hipSuccess && hipStreamWaitEvent(l->side, l->fork, 0) == hipSuccess) {
      // (the order of the two launches and the side stream's priority change nothing measurable)
      int rc = launch_ragged_interior<B_KN>(A, B, C, M, N, K, swizzle_stride, st);
      if (rc == LC_OK) rc = launch_hgemm_mid_edge(A, B, C, M, N, K, B_KN, 2, 2, p.ns, p.Mi, p.Ni, l->side);
      const bool joined = hipEventRecord(l->join, l->side) == hipSuccess;
      if (!joined || hipStreamWaitEvent(st, l->join, 0) != hipSuccess) {   // (cannot order the caller's stream behind the border: wait for it here)
        (void)hipGetLastError();
        (void)hipStreamSynchronize(l->side);
      }
      return rc;
    }
    (void)hipGetLastError();
  }
  if (int rc = launch_ragged_interior<B_KN>(A, B, C, M, N, K, swizzle_stride, st)) return rc;
  return launch_hgemm_mid_edge(A, B, C, M, N, K, B_KN, 2, 2, p.ns, p.Mi, p.Ni, st);
}
}  // namespace

// ------------------------------------------------------------------------------------------------
// vendor comparator (hipBLASLt), resolved lazily with dlopen so the core library has no link-time
// dependency on it.

#include "vendor_gemm.inc"

extern "C" {

int lc_abi_version(void) { return LC_ABI_VERSION; }

const char* lc_status_string(int status) {
  switch (status) {
    case LC_OK: return "ok";
    case LC_ERR_ARG: return "invalid argument";
    case LC_ERR_SHAPE: return "Tensor size mismatch!";
    case LC_ERR_HEADDIM: return "headdim not support!";
    case LC_ERR_LAUNCH: return "kernel launch failed";
    case LC_ERR_VENDOR: return "vendor GEMM (hipBLASLt) unavailable or failed";
    case LC_ERR_DEVICE: return "no gfx950 device";
    default: return "unknown status";
  }
}

const char* lc_build_info(int* is_diag) {
#ifdef LC_DIAG
  if (is_diag) *is_diag = 1;
  return "gfx950 -O3 LC_DIAG=1 (diagnosis build: ablation / stamp kernels compiled in)";
#else
  if (is_diag) *is_diag = 0;
  return "gfx950 -O3 LC_DIAG=0";
#endif
}

namespace {
bool is_tile256_variant(int v) { return v == LC_HGEMM_MFMA256 || v == LC_HGEMM_MFMA256P2 || is_w4_variant(v); }
bool is_valu_variant(int v) { return v >= LC_HGEMM_VALU_NAIVE && v <= LC_HGEMM_VALU_T16X8_K32; }
bool is_hgemm_variant(int v) {
  return v == LC_HGEMM_AUTO || v == LC_HGEMM_GENERIC || v == LC_HGEMM_EDGE || v == LC_HGEMM_RAGGED || v == LC_HGEMM_KPAD || v == LC_HGEMM_MFMA128 || v == LC_HGEMM_MID || is_tile256_variant(v) || is_valu_variant(v);
}
}  // namespace

namespace {
// LC_HGEMM_AUTO -> a concrete kernel family; tile checks of the explicit families.  ONE function for lc_hgemm_f16
// and lc_hgemm_kernel_name().  Returns the variant or a negative lc_status.
// Shapes (the reference's kernels are legal on M, N multiples of 128 and K multiples of 32, hgemm_mma_stage.cu:650,675-676):
//   hgemm_w4y_kernel        M, N % 128 == 0 with a 256-tileable interior (the 128-wide border strips run on the 128-tile kernel,
//                           launch_mfma256), K % 32 == 0, K >= 64 (K % 64 == 32: a half K-step behind the generated loop)
//   hgemm_mfma128_kernel    M, N % 128 == 0, K % 32 == 0, K >= 64
//   the other 256-tile kernels (cross-checks): M, N % 256 == 0, K % 64 == 0
int resolve_hgemm_variant(int variant, int M, int N, int K, bool al, bool b_kn) {
  const bool k64 = K % BK == 0, k32 = K % 32 == 0 && K >= BK;
  const bool tiles256 = (M % BM == 0) && (N % BN == 0) && k64 && al;
  const bool tiles128 = (M % BM1 == 0) && (N % BN1 == 0) && k32 && al;
  const bool edge_ok = al && K % 8 == 0 && (!b_kn || N % 8 == 0);   // hgemm_edge_kernel: whole 16-byte chunks
  // hgemm_w4y_kernel itself (not the 64-bit-address kernel w4_effective_variant substitutes for huge operands) on this shape
  const bool w4y_ok = tiles128 && M >= BM && N >= BN && w4_effective_variant(LC_HGEMM_MFMA256W4Y, b_kn, N, K) == LC_HGEMM_MFMA256W4Y;
  if (variant == LC_HGEMM_AUTO) {
    // measured crossover on MI355X (TN, square): the 256-tile kernel wins once its grid has more
    // than ~128 workgroups (n >= 3072); below that the 128-tile kernel fills the 256 CUs better
    // (n = 2048: 715 vs 436 TFLOP/s).
    const long wg256 = (long)(M / BM) * (N / BN);
    const int a = g_tune_hgemm_auto;
    const bool tiles64 = (M % 64 == 0) && (N % 64 == 0) && k32 && al;
    if (2 * wg256 > rule_cu_count() && (tiles256 || (a == LC_HGEMM_MFMA256W4Y && w4y_ok))) {   // more than half a CU's worth of 256 x 256 tiles per CU (256 CUs: > 128)
      // ... unless those tiles leave CUs idle in their ONE round and a mid-size tile fills more of them in one round of its own
      // (3072^3 TN: 144 tiles of 256 x 256 against 256 of 192 x 192, 1050 -> 1110 TFLOP/s, profiles/r6p_hgemm_rect_splitk.log)
      if (wg256 < rule_cu_count()) {
        const MidTile t = mid_tile_auto(M, N, K, b_kn, true);
        if (t.tmw > 0 && t.wgs > wg256) return LC_HGEMM_MID;
      }
      return a;
    }
    // ragged M / N whose interior fills the flagship kernel: that kernel + a border launch, ahead of a 64-multiple tile of the mid-size kernel
    // (8192 x 8256 x 4096 TN: 1261 against 1096 TFLOP/s on 128 x 192 tiles, profiles/r6ab_hgemm_edge_ab.log)
    const int rk = tiles128 ? 0 : ragged_plan(M, N, K, al, b_kn, true).kind;
    if (rk == 1) return LC_HGEMM_RAGGED;
    if (tiles64 && mid_tile_auto(M, N, K, b_kn, true).tmw > 0) return LC_HGEMM_MID;   // the tile with the least work on the busiest CU (n = 1280 .. 2816 square)
    if (tiles128) return LC_HGEMM_MFMA128;
    if (rk) return LC_HGEMM_RAGGED;   // the whole problem on 128 x 128 tiles of the mid-size kernel that may reach beyond M / N
    if (kpad_plan(M, N, K, al, true)) return LC_HGEMM_KPAD;   // K % 32 != 0 on a large problem: zero-padded operand copies + the tuned kernels
    return edge_ok ? LC_HGEMM_EDGE : LC_HGEMM_GENERIC;
  }
  if (is_valu_variant(variant)) {   // a rung of the vector-ALU ladder: its own tile, else the edge kernel (never an error)
    int tm, tn, tk;
    valu_rung_tile(variant, &tm, &tn, &tk);
    const bool ok = (M % tm == 0) && (N % tn == 0) && (K % tk == 0) && (variant == LC_HGEMM_VALU_NAIVE || (al && K % 8 == 0));
    return ok ? variant : LC_HGEMM_GENERIC;
  }
  if (variant == LC_HGEMM_MFMA256W4Y && w4y_ok) return variant;
  if (is_tile256_variant(variant) && !tiles256) return LC_ERR_SHAPE;
  if (variant == LC_HGEMM_MFMA128 && !tiles128) return LC_ERR_SHAPE;
  if (variant == LC_HGEMM_MID && !(al && mid_tile_auto(M, N, K, b_kn, false).tmw > 0)) return LC_ERR_SHAPE;
  if (variant == LC_HGEMM_EDGE && !edge_ok) return LC_ERR_SHAPE;
  if (variant == LC_HGEMM_RAGGED && !ragged_plan(M, N, K, al, b_kn, false).kind) return LC_ERR_SHAPE;
  if (variant == LC_HGEMM_KPAD && !kpad_plan(M, N, K, al, false)) return LC_ERR_SHAPE;
  return variant;
}
}  // namespace

int lc_hgemm_kernel_name(int M, int N, int K, int layout, int variant, char* buf, int buflen) {
  if (!buf || buflen < 8 || M <= 0 || N <= 0 || K <= 0 || !is_hgemm_variant(variant)) return LC_ERR_ARG;
  if (layout != LC_LAYOUT_NN && layout != LC_LAYOUT_TN) return LC_ERR_ARG;
  int v = resolve_hgemm_variant(variant, M, N, K, true, layout == LC_LAYOUT_NN);
  if (v < 0) return v;
  if (is_valu_variant(v) && layout != LC_LAYOUT_NN) v = LC_HGEMM_GENERIC;   // the ladder is NN only
  const char* nn = layout == LC_LAYOUT_NN ? "true" : "false";
  if (is_valu_variant(v)) {
    snprintf(buf, buflen, "%s", valu_rung_kernel_name(v));
    return LC_OK;
  }
  if (is_w4_variant(v)) {
    v = w4_effective_variant(v, layout == LC_LAYOUT_NN, N, K);
    if (v == LC_HGEMM_MFMA256W4X) snprintf(buf, buflen, "hgemm_w4x_kernel<%s>", nn);
    else if (v == LC_HGEMM_MFMA256W4Y) snprintf(buf, buflen, "hgemm_w4y_kernel<%s,%d>", nn, layout == LC_LAYOUT_NN ? 1 : g_tune_w4y_sched.load());
    else snprintf(buf, buflen, "hgemm_w4b_kernel<%s,%s,0>", nn, v == LC_HGEMM_MFMA256W4B ? "false" : "true");
  } else if (v == LC_HGEMM_MFMA256P2) snprintf(buf, buflen, "hgemm_pingpong2_kernel<%s,false>", nn);
  else if (v == LC_HGEMM_MFMA256) snprintf(buf, buflen, "hgemm_mfma256_kernel<%s>", nn);
  else if (v == LC_HGEMM_MID) {
    const MidTile t = mid_tile_auto(M, N, K, layout == LC_LAYOUT_NN, variant != LC_HGEMM_MID);
    if (t.ks > 1) snprintf(buf, buflen, "hgemm_mid_sk_kernel<%s,%d,%d> x%d", nn, t.tmw, t.ns, t.ks);   // (x K ranges, + hgemm_mid_reduce_kernel; hgemm_mid_kernel under graph capture)
    else snprintf(buf, buflen, "hgemm_mid_kernel<%s,%d,%d,%d>", nn, t.tmw, t.tnw, t.ns);
  } else if (v == LC_HGEMM_MFMA128) snprintf(buf, buflen, "hgemm_mfma128_kernel<%s,%d>", nn, mfma128_ksw((long)(M / BM1) * (N / BN1)));
  else if (v == LC_HGEMM_EDGE) snprintf(buf, buflen, "hgemm_edge_kernel<%s>", nn);
  else if (v == LC_HGEMM_KPAD) {   // the copies + whatever the padded problem runs
    const int Kp = kpad_plan(M, N, K, true, variant != LC_HGEMM_KPAD);
    char inner[160];
    if (Kp <= 0 || lc_hgemm_kernel_name(M, N, Kp, layout, LC_HGEMM_AUTO, inner, (int)sizeof(inner)) != LC_OK) snprintf(buf, buflen, "hgemm_edge_kernel<%s>", nn);
    else snprintf(buf, buflen, "hgemm_pad_copy_kernel + %s", inner);
  } else if (v == LC_HGEMM_RAGGED) {   // interior kernel + the border launch
    const RaggedPlan p = ragged_plan(M, N, K, true, layout == LC_LAYOUT_NN, variant != LC_HGEMM_RAGGED);
    if (p.kind == 1) snprintf(buf, buflen, "hgemm_w4y_kernel<%s,%d> + hgemm_mid_edge_kernel<%s,2,2,%d>", nn, layout == LC_LAYOUT_NN ? 1 : g_tune_w4y_sched.load(), nn, p.ns);
    else if (p.kind == 2 && p.ks > 1) snprintf(buf, buflen, "hgemm_mid_edge_sk_kernel<%s,%d,3> x%d", nn, p.tmw, p.ks);   // (x K ranges, + hgemm_mid_reduce_edge_kernel; hgemm_mid_edge_kernel under graph capture)
    else if (p.kind == 2) snprintf(buf, buflen, "hgemm_mid_edge_kernel<%s,%d,%d,%d>", nn, p.tmw, p.tnw, p.ns);
    else snprintf(buf, buflen, "hgemm_edge_kernel<%s>", nn);   // (the knob changed between the two reads)
  } else snprintf(buf, buflen, "hgemm_generic_kernel<%s>", nn);
  return LC_OK;
}

int lc_attn_kernel_name(int N, int D, int v_transposed, int bf16, char* buf, int buflen) {
  return lc_attn_kernel_name_bh(-1, N, D, v_transposed, bf16, buf, buflen);
}

int lc_attn_kernel_name_bh(int BH, int N, int D, int v_transposed, int bf16, char* buf, int buflen) {
  if (!buf || buflen < 8 || N <= 0 || N % KVB != 0) return LC_ERR_ARG;
  if (D > 0 && (size_t)N * (size_t)D * 2 >= 0x80000000ull) return LC_ERR_SHAPE;   // (mirrors lc_attn_fwd_f16)
  const char* vt = v_transposed ? "true" : "false";
  if (D == 32 || D == 64 || D == 96 || D == 128) {
    if (bf16) return LC_ERR_HEADDIM;
    const int nw = choose_attn_nw(D, v_transposed != 0, N, BH > 0 ? (long)BH : -1);
    // (a persistent walk with no more blocks than CUs launches WALK 0; the name reports the walk asked for at this N; 3 = split-KV, whose
    // launch also runs attn_split_combine_kernel<D>)
    if (nw == 513 || nw == 515 || nw == 517 || nw == 519) snprintf(buf, buflen, "attn_fwd_w4u_kernel<%d,%s,%d>", D, vt, (nw - 513) / 2);
    else if (nw == 514) snprintf(buf, buflen, "attn_fwd_w4i_kernel<%d,%d>", D, g_tune_attn_w4i_sched.load());
    else snprintf(buf, buflen, "attn_fwd_kernel<%d,%d,%s,0>", D, nw, vt);
    return LC_OK;
  }
  if (use_bigd4(D, v_transposed != 0, N) && !bf16) {
    snprintf(buf, buflen, "attn_fwd_bigd4_kernel<%d>", g_tune_attn_d1024 == 0 ? 8 : g_tune_attn_d1024.load());
    return LC_OK;
  }
  if (use_bigd6(D, v_transposed != 0, N)) {
    snprintf(buf, buflen, "attn_fwd_bigd6_kernel<%s>", bf16 ? "true" : "false");
    return LC_OK;
  }
  if (use_bigd7(D, v_transposed != 0, N, BH > 0 ? (long)BH : -1) && !(bf16 && v_transposed)) {
    snprintf(buf, buflen, "attn_fwd_bigd7_kernel<%s,%s>", bf16 ? "true" : "false", v_transposed ? "true" : "false");
    return LC_OK;
  }
  if (use_bigd2(D, v_transposed != 0, N) && !(bf16 && v_transposed)) {
    if (v_transposed) snprintf(buf, buflen, "attn_fwd_bigd2_kernel<%d,false,true>", D);
    else if (g_tune_attn_d512 == 2) snprintf(buf, buflen, "attn_fwd_bigd3_kernel<%d,%s>", D, bf16 ? "true" : "false");
    else snprintf(buf, buflen, "attn_fwd_bigd2_kernel<%d,%s,false>", D, bf16 ? "true" : "false");
    return LC_OK;
  }
  if (D == 256 || D == 512 || (D == 1024 && !bf16)) {
    snprintf(buf, buflen, "attn_fwd_bigd_kernel<%d,%d,%d,%s,%s>", D, D > 256 ? 256 : D, N % 128 == 0 ? 4 : 2, vt,
             bf16 ? "true" : "false");
    return LC_OK;
  }
  return LC_ERR_HEADDIM;
}

namespace {
// the knob registry: ONE table for lc_tune_set / lc_tune_get (key, variable, default, validity of a value)
bool ok_attn_nw(int v) {
  return v == 0 || v == 512 || v == 513 || v == 514 || v == 515 || v == 517 || v == 8 || v == 4 || v == 2;
}
bool ok_01(int v) { return v == 0 || v == 1; }
bool ok_02(int v) { return v >= 0 && v <= 2; }
bool ok_03(int v) { return v >= 0 && v <= 3; }
bool ok_04(int v) { return v >= 0 && v <= 4; }
bool ok_08(int v) { return v >= 0 && v <= 8; }
bool ok_rule_cus(int v) { return v == 0 || (v >= 64 && v <= 1024); }
bool ok_mid_ns(int v) { return v == 0 || v == 2 || v == 3; }
bool ok_ragged_tile(int v) { return v == 0 || v == 12 || v == 22 || v == 23 || v == 32 || v == 33; }
bool ok_mid(int v) { return v == 0 || v == 1 || v == 12 || v == 13 || v == 22 || v == 23 || v == 32 || v == 33; }
bool ok_split(int v) { return v == 0 || v == 1 || v == 2 || v == 4 || v == 8 || v == 16; }
bool ok_span8(int v) { return v == 0 || v == 2 || v == 4 || v == 6; }
bool ok_w4y_sched(int v) {
#ifdef LC_DIAG
  return v >= 0 && v <= 5;   // 3..5: ablations (results WRONG)
#else
  return v >= 0 && v <= 2;
#endif
}
// cx | cm << 4 | cn << 8 | step << 12 | mask << 20 with a 7-bit mask (bits 20 .. 26), or exactly STAGGER_OFF (1 << 27)
bool ok_stagger(int v) { return v >= 0 && ((v >> 27) == 0 || v == STAGGER_OFF); }
bool ok_auto(int v) { return is_tile256_variant(v); }
bool ok_any(int) { return true; }
struct Knob {
  const char* key;
  tune_t* var;
  int dflt;
  bool (*valid)(int);
  bool diag;   // diagnosis key (include/lc_diag.h): results may be WRONG; a production library rejects it
};
const Knob kKnobs[] = {
    {"attn_nw", &g_tune_attn_nw, 0, ok_attn_nw, false},
    {"attn_walk", &g_tune_attn_walk, 0, ok_03, false},
    {"attn_split", &g_tune_attn_split, 0, ok_split, false},
    {"attn_bigd_map", &g_tune_attn_bigd_map, 0, ok_02, false},
    {"attn_bigd_stagger", &g_tune_attn_bigd_stagger, 0, ok_02, false},
    {"attn_d1024", &g_tune_attn_d1024, 0, ok_span8, false},
    {"attn_w4i_sched", &g_tune_attn_w4i_sched, 1, ok_01, false},
    {"fp8_mx", &g_tune_fp8_mx, 3, ok_03, false},
    {"attn_d512", &g_tune_attn_d512, 0, ok_04, false},
    {"w4y_sched", &g_tune_w4y_sched, 2, ok_w4y_sched, false},
    {"hgemm_persist", &g_tune_hgemm_persist, 1, ok_01, false},
    {"hgemm_stagger", &g_tune_hgemm_stagger, 0, ok_stagger, false},
    {"hgemm_tail", &g_tune_hgemm_tail, 1, ok_04, false},
    {"hgemm_tail_tile", &g_tune_hgemm_tail_tile, 0, ok_02, false},
    {"hgemm_ragged", &g_tune_hgemm_ragged, 0, ok_01, false},
    {"hgemm_ragged_fork", &g_tune_hgemm_ragged_fork, 0, ok_02, false},
    {"hgemm_ragged_tile", &g_tune_hgemm_ragged_tile, 0, ok_ragged_tile, false},
    {"hgemm_kpad", &g_tune_hgemm_kpad, 0, ok_02, false},
    {"hgemm_mid_splitk", &g_tune_hgemm_mid_splitk, 0, ok_08, false},
    {"hgemm_128w", &g_tune_hgemm_128w, 0, ok_02, false},
    {"rule_cus", &g_tune_rule_cus, 0, ok_rule_cus, false},
    {"attn_calib", &g_tune_attn_calib, 0, ok_01, false},
    {"hgemm_mid", &g_tune_hgemm_mid, 0, ok_mid, false},
    {"hgemm_mid_ns", &g_tune_hgemm_mid_ns, 0, ok_mid_ns, false},
    {"hgemm_splitk", &g_tune_hgemm_splitk, 0, ok_08, false},
    {"hgemm_raster", &g_tune_hgemm_raster, 0, ok_02, false},
    {"hgemm_auto", &g_tune_hgemm_auto, LC_HGEMM_MFMA256W4Y, ok_auto, false},
    {"w4_abl", &g_tune_w4_abl, 0, ok_any, true},
    {"hgemm_stamps", &g_tune_hgemm_stamps, 0, ok_01, true},
    {"attn_ablate", &g_tune_attn_ablate, 0, ok_any, true},
};
const Knob* find_knob(const char* key) {
  if (!key) return nullptr;
  for (const Knob& k : kKnobs)
    if (strcmp(k.key, key) == 0) {
#ifndef LC_DIAG
      if (k.diag) return nullptr;
#endif
      return &k;
    }
  return nullptr;
}
}  // namespace

int lc_tune_set(const char* key, int value) {
  const Knob* k = find_knob(key);
  if (!k || !k->valid(value)) return LC_ERR_ARG;
  k->var->store(value, std::memory_order_relaxed);
  return LC_OK;
}

int lc_tune_get(const char* key, int* value, int* default_value) {
  const Knob* k = find_knob(key);
  if (!k) return LC_ERR_ARG;
  if (value) *value = k->var->load(std::memory_order_relaxed);
  if (default_value) *default_value = k->dflt;
  return LC_OK;
}

int lc_tune_count(void) { return (int)(sizeof(kKnobs) / sizeof(kKnobs[0])); }
size_t lc_workspace_release(void) { return workspace_release_all(); }
size_t lc_workspace_bytes(void) { return workspace_cached_bytes(); }
const char* lc_tune_key(int index) {
  if (index < 0 || index >= lc_tune_count()) return nullptr;
#ifndef LC_DIAG
  if (kKnobs[index].diag) return nullptr;
#endif
  return kKnobs[index].key;
}

int lc_device_check(int* num_cus) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return LC_ERR_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return LC_ERR_DEVICE;
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return LC_ERR_DEVICE;
  if (num_cus) *num_cus = prop.multiProcessorCount;
  return LC_OK;
}

int lc_hgemm_f16(const void* A, const void* B, void* C, int M, int N, int K, int layout, int variant,
                 int stages, int swizzle_stride, void* stream) {
  (void)stages;  // accepted and ignored: the LDS ring depth is fixed per kernel family (lc_abi.h)
  if (!A || !B || !C) return LC_ERR_ARG;
  if (layout != LC_LAYOUT_NN && layout != LC_LAYOUT_TN) return LC_ERR_ARG;
  if (!is_hgemm_variant(variant)) return LC_ERR_ARG;
  if (M <= 0 || N <= 0 || K <= 0) return LC_ERR_SHAPE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const half_t* a = static_cast<const half_t*>(A);
  const half_t* b = static_cast<const half_t*>(B);
  half_t* c = static_cast<half_t*>(C);
  const bool al = aligned16(A) && aligned16(B) && aligned16(C);
  const bool mid_forced = variant == LC_HGEMM_MID, ragged_forced = variant == LC_HGEMM_RAGGED, kpad_forced = variant == LC_HGEMM_KPAD;
  variant = resolve_hgemm_variant(variant, M, N, K, al, layout == LC_LAYOUT_NN);
  if (variant < 0) return variant;
  if (int rc = launch_guard()) return rc;   // a sticky HIP error of an earlier call: report it, launch nothing
  if (variant == LC_HGEMM_MID) {
    const MidTile t = mid_tile_auto(M, N, K, layout == LC_LAYOUT_NN, !mid_forced);
    if (t.tmw > 0) return launch_mid(a, b, c, M, N, K, layout == LC_LAYOUT_NN, t, swizzle_stride, st);
    if (!(M % BM1 == 0 && N % BN1 == 0)) return layout == LC_LAYOUT_NN ? launch_edge<true>(a, b, c, M, N, K, 0, 0, st) : launch_edge<false>(a, b, c, M, N, K, 0, 0, st);
    variant = LC_HGEMM_MFMA128;   // (the knob changed between the two reads; every LC_HGEMM_MID shape is an edge-kernel shape)
  }
  if (is_valu_variant(variant)) {
    if (layout == LC_LAYOUT_NN) return launch_valu_rung(a, b, c, M, N, K, variant, st);
    variant = LC_HGEMM_GENERIC;   // the ladder is NN only
  }
  if (is_tile256_variant(variant)) {
    return layout == LC_LAYOUT_NN ? launch_mfma256<true>(a, b, c, M, N, K, variant, swizzle_stride, st)
                                  : launch_mfma256<false>(a, b, c, M, N, K, variant, swizzle_stride, st);
  }
  if (variant == LC_HGEMM_MFMA128) {
    return layout == LC_LAYOUT_NN ? launch_mfma128<true>(a, b, c, M, N, K, swizzle_stride, st)
                                  : launch_mfma128<false>(a, b, c, M, N, K, swizzle_stride, st);
  }
  if (variant == LC_HGEMM_KPAD) {
    const int Kp = kpad_plan(M, N, K, al, !kpad_forced);
    if (Kp > 0 && !workspace_held_by_this_thread() && !stream_is_capturing(st)) {
      WorkspaceLease lease = stream_workspace(st, ((size_t)M + N) * Kp * 2);
      if (lease.ptr) {
        half_t* ap = static_cast<half_t*>(lease.ptr);
        half_t* bp = ap + (size_t)M * Kp;
        const bool nn = layout == LC_LAYOUT_NN;
        // A [M][K] -> [M][Kp]; B as [N][K] -> [N][Kp], as [K][N] -> [Kp][N] (zero rows behind the last k)
        const size_t ca = (size_t)M * (Kp / 8), cb = nn ? (size_t)Kp * (N / 8) : (size_t)N * (Kp / 8);
        hipLaunchKernelGGL(hgemm_pad_copy_kernel, dim3((unsigned)((ca + 255) / 256)), dim3(256), 0, st, a, ap, M, K, M, Kp);
        if (nn) hipLaunchKernelGGL(hgemm_pad_copy_kernel, dim3((unsigned)((cb + 255) / 256)), dim3(256), 0, st, b, bp, K, N, Kp, N);
        else hipLaunchKernelGGL(hgemm_pad_copy_kernel, dim3((unsigned)((cb + 255) / 256)), dim3(256), 0, st, b, bp, N, K, N, Kp);
        if (int rc = check_launch()) return rc;
        workspace_held_by_this_thread() = true;   // the padded problem's launch: workspace-free forms, no fork
        const int rc = lc_hgemm_f16(ap, bp, C, M, N, Kp, layout, LC_HGEMM_AUTO, stages, swizzle_stride, stream);
        workspace_held_by_this_thread() = false;
        return rc;
      }
    }
    variant = LC_HGEMM_EDGE;   // (graph capture, no workspace, the knob changed between the two reads: every LC_HGEMM_KPAD shape is an edge-kernel shape)
  }
  if (variant == LC_HGEMM_RAGGED) {
    const RaggedPlan p = ragged_plan(M, N, K, al, layout == LC_LAYOUT_NN, !ragged_forced);
    if (p.kind) return layout == LC_LAYOUT_NN ? launch_ragged<true>(a, b, c, M, N, K, p, swizzle_stride, st)
                                              : launch_ragged<false>(a, b, c, M, N, K, p, swizzle_stride, st);
    variant = LC_HGEMM_EDGE;   // (the knob changed between the two reads; every LC_HGEMM_RAGGED shape is an edge-kernel shape)
  }
  if (variant == LC_HGEMM_EDGE) {
    return layout == LC_LAYOUT_NN ? launch_edge<true>(a, b, c, M, N, K, 0, 0, st) : launch_edge<false>(a, b, c, M, N, K, 0, 0, st);
  }
  return layout == LC_LAYOUT_NN ? launch_generic<true>(a, b, c, M, N, K, st)
                                : launch_generic<false>(a, b, c, M, N, K, st);
}

int lc_gemm_fp8_e4m3(const void* A, const void* B, void* C, int M, int N, int K, float alpha,
                     int swizzle_stride, void* stream) {
  if (!A || !B || !C) return LC_ERR_ARG;
  if (M <= 0 || N <= 0 || K <= 0) return LC_ERR_SHAPE;
  if (M % BM || N % BN || K % 128 || !aligned16(A) || !aligned16(B) || !aligned16(C)) return LC_ERR_SHAPE;
  if (int rc = launch_guard()) return rc;
  const int tiles_m = M / BM, tiles_n = N / BN;
  const int pw = panel_tiles(swizzle_stride, tiles_n, BN, ((size_t)M + N) * K);   // (fp8: one byte per element)
  const int mx = g_tune_fp8_mx;   // read once per launch
  if (mx == 3 && gemm_fp8_w4k_fits(K))
    return launch_gemm_fp8_w4k(static_cast<const uint8_t*>(A), static_cast<const uint8_t*>(B), static_cast<half_t*>(C), M, N, K, alpha,
                               tiles_m, tiles_n, pw, static_cast<hipStream_t>(stream));
  return launch_gemm_fp8(static_cast<const uint8_t*>(A), static_cast<const uint8_t*>(B), static_cast<half_t*>(C), M, N, K,
                         alpha, tiles_m, tiles_n, pw, mx == 3 ? 1 : mx, static_cast<hipStream_t>(stream));
}

int lc_mxfp8_pack_scales(const void* S, void* P, int rows, int K, void* stream) {
  if (!S || !P) return LC_ERR_ARG;
  if (rows <= 0 || K <= 0) return LC_ERR_SHAPE;
  if (rows % BM || K % 128 || ((uintptr_t)P & 7)) return LC_ERR_SHAPE;
  if (int rc = launch_guard()) return rc;
  return launch_mx_pack_scales(static_cast<const uint8_t*>(S), static_cast<uint32_t*>(P), rows, K, static_cast<hipStream_t>(stream));
}

int lc_gemm_mxfp8(const void* A, const void* PA, const void* B, const void* PB, void* C, int M, int N, int K, float alpha,
                  int swizzle_stride, void* stream) {
  if (!A || !B || !C || !PA || !PB) return LC_ERR_ARG;
  if (M <= 0 || N <= 0 || K <= 0) return LC_ERR_SHAPE;
  if (M % BM || N % BN || K % 128 || !aligned16(A) || !aligned16(B) || !aligned16(C) || ((uintptr_t)PA & 7) || ((uintptr_t)PB & 7))
    return LC_ERR_SHAPE;
  if (!gemm_fp8_w4k_fits(K)) return LC_ERR_SHAPE;   // 32-bit DMA offsets (K < 8 Mi)
  if (int rc = launch_guard()) return rc;
  const int tiles_m = M / BM, tiles_n = N / BN;
  const int pw = panel_tiles(swizzle_stride, tiles_n, BN, ((size_t)M + N) * K);
  return launch_gemm_mxfp8(static_cast<const uint8_t*>(A), static_cast<const uint32_t*>(PA), static_cast<const uint8_t*>(B),
                           static_cast<const uint32_t*>(PB), static_cast<half_t*>(C), M, N, K, alpha, tiles_m, tiles_n, pw,
                           static_cast<hipStream_t>(stream));
}

int lc_hgemm_entry_count(void) { return kNumHgemmEntries; }
const char* lc_hgemm_entry_name(int index) {
  return (index >= 0 && index < kNumHgemmEntries) ? kHgemmEntries[index].name : nullptr;
}
int lc_hgemm_entry_info(const char* entry, int* layout, int* nargs) {
  const HgemmEntry* e = find_hgemm(entry);
  if (!e) return LC_ERR_ARG;
  if (layout) *layout = e->layout;
  if (nargs) *nargs = e->nargs;
  return LC_OK;
}

int lc_hgemm_call(const char* entry, const void* A, const void* B, void* C, int M, int N, int K,
                  int stages, int swizzle, int swizzle_stride, void* stream) {
  const HgemmEntry* e = find_hgemm(entry);
  if (!e) return LC_ERR_ARG;
  if (e->variant == -2) return lc_vendor_init();
  if (e->variant == -3) return lc_vendor_destroy();
  if (e->variant == -1) return lc_hgemm_vendor_f16(A, B, C, M, N, K, e->layout, stream);
  int variant = e->variant;
  // the cross-check kernels need 256-multiples and K % 64 == 0; every reference entry must still accept the reference's own
  // legal shapes (multiples of 128 / K of 32, hgemm_mma_stage.cu:650,675), so fall back per shape: AUTO serves those with the
  // flagship kernel + border strips, the 128-tile kernel or the edge kernel (resolve_hgemm_variant)
  const bool al = aligned16(A) && aligned16(B) && aligned16(C);
  const bool tiles256 = (M % BM == 0) && (N % BN == 0) && (K % BK == 0) && al;
  const bool tiles128 = (M % BM1 == 0) && (N % BN1 == 0) && (K % 32 == 0) && K >= BK && al;
  if (is_tile256_variant(variant) && !tiles256) variant = LC_HGEMM_AUTO;
  if (variant == LC_HGEMM_MFMA128 && !tiles128) variant = LC_HGEMM_GENERIC;
  // Round 6: the entry names that map to the cross-check kernels keep them where those kernels are within a few per cent of the flagship
  // (large grids: the reference bench's rows stay distinct there), but a 256 x 256 tile on a grid of at most half a CU's worth of tiles
  // per CU — 2048^3: 64 workgroups, 330 TFLOP/s where LC_HGEMM_AUTO reaches 818 in the reference's own unmodified sweep
  // (profiles/r6Z_f1_hgemm_default_sweep.log) — serves nobody: such calls, and 128-tile names on shapes the mid-size kernel serves, run
  // what LC_HGEMM_AUTO runs.
  if (is_tile256_variant(variant) && variant != LC_HGEMM_MFMA256W4Y && 2L * (M / BM) * (N / BN) <= rule_cu_count()) variant = LC_HGEMM_AUTO;
  if (variant == LC_HGEMM_MFMA128 && resolve_hgemm_variant(LC_HGEMM_AUTO, M, N, K, al, e->layout == LC_LAYOUT_NN) == LC_HGEMM_MID) variant = LC_HGEMM_AUTO;
  const int stride = (e->nargs == 6 && swizzle) ? swizzle_stride : 1;
  return lc_hgemm_f16(A, B, C, M, N, K, e->layout, variant, e->nargs == 6 ? stages : 2, stride, stream);
}

int lc_attn_fwd_f16(const void* Q, const void* K, const void* V, void* O, int B, int H, int N, int D,
                    int v_transposed, int family, int acc_f32, int stages, void* stream) {
  (void)acc_f32;
  (void)stages;
  if (!Q || !K || !V || !O) return LC_ERR_ARG;
  if (family < LC_ATTN_SPLIT_Q || family > LC_ATTN_SPLIT_KV) return LC_ERR_ARG;
  if (B <= 0 || H <= 0 || N <= 0 || D <= 0) return LC_ERR_SHAPE;
  if (N % KVB != 0) return LC_ERR_SHAPE;
  if ((size_t)B * H * (size_t)(N / 64) > 0x7fffffffull) return LC_ERR_SHAPE;  // 1-D grid of workgroups
  if ((size_t)N * (size_t)D * 2 >= 0x80000000ull) return LC_ERR_SHAPE;   // one head's K / V must fit the 32-bit buffer offsets of the LDS-DMA kernels
  if (!aligned16(Q) || !aligned16(K) || !aligned16(V) || !aligned16(O)) return LC_ERR_SHAPE;
  if (int rc = launch_guard()) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const half_t* q = static_cast<const half_t*>(Q);
  const half_t* k = static_cast<const half_t*>(K);
  const half_t* v = static_cast<const half_t*>(V);
  half_t* o = static_cast<half_t*>(O);
  return v_transposed ? launch_attn_d<true>(q, k, v, o, B, H, N, D, st)
                      : launch_attn_d<false>(q, k, v, o, B, H, N, D, st);
}

int lc_attn_fwd_bf16(const void* Q, const void* K, const void* V, void* O, int B, int H, int N, int D,
                     void* stream) {
  if (!Q || !K || !V || !O) return LC_ERR_ARG;
  if (B <= 0 || H <= 0 || N <= 0 || D <= 0 || N % KVB != 0) return LC_ERR_SHAPE;
  if ((size_t)B * H * (size_t)(N / 64) * 4 > 0x7fffffffull) return LC_ERR_SHAPE;
  if ((size_t)N * (size_t)D * 2 >= 0x80000000ull) return LC_ERR_SHAPE;   // 32-bit buffer offsets inside one head (as lc_attn_fwd_f16)
  if (!aligned16(Q) || !aligned16(K) || !aligned16(V) || !aligned16(O)) return LC_ERR_SHAPE;
  if (int rc = launch_guard()) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const half_t* q = static_cast<const half_t*>(Q);   // raw 16-bit lanes; the kernel flavour decodes bf16
  const half_t* k = static_cast<const half_t*>(K);
  const half_t* v = static_cast<const half_t*>(V);
  half_t* o = static_cast<half_t*>(O);
  if (use_bigd6(D, false, N)) return launch_attn_bigd6(q, k, v, o, B, H, N, true, st);
  if (use_bigd7(D, false, N, (long)B * H)) return launch_attn_bigd7(q, k, v, o, B, H, N, true, st);
  if (use_bigd2(D, false, N)) return launch_attn_bigd2(q, k, v, o, B, H, N, D, true, st);
  const bool w4 = N % 128 == 0;
  switch (D) {
    case 256:
      return w4 ? launch_attn_bigd<256, 4, false, true>(q, k, v, o, B, H, N, st)
                : launch_attn_bigd<256, 2, false, true>(q, k, v, o, B, H, N, st);
    case 512:
      return w4 ? launch_attn_bigd<512, 4, false, true>(q, k, v, o, B, H, N, st)
                : launch_attn_bigd<512, 2, false, true>(q, k, v, o, B, H, N, st);
    default: return LC_ERR_HEADDIM;
  }
}

int lc_attn_entry_count(void) { return kNumAttnEntries; }
const char* lc_attn_entry_name(int index) {
  return (index >= 0 && index < kNumAttnEntries) ? kAttnEntries[index].name : nullptr;
}
int lc_attn_entry_info(const char* entry, int* family, int* v_transposed, int* acc_f32,
                       int* max_d_stage2, int* max_d_stage1, int* nargs) {
  const AttnEntry* e = find_attn(entry);
  if (!e) return LC_ERR_ARG;
  if (family) *family = e->family;
  if (v_transposed) *v_transposed = e->vt;
  if (acc_f32) *acc_f32 = e->acc_f32;
  if (max_d_stage2) *max_d_stage2 = e->maxd_s2;
  if (max_d_stage1) *max_d_stage1 = e->maxd_s1;
  if (nargs) *nargs = e->nargs;
  return LC_OK;
}

int lc_attn_call(const char* entry, const void* Q, const void* K, const void* V, void* O, int B, int H,
                 int N, int D, int stages, void* stream) {
  const AttnEntry* e = find_attn(entry);
  if (!e) return LC_ERR_ARG;
  const int st2 = (e->nargs == 4) ? 2 : stages;
  const int maxd = st2 > 1 ? e->maxd_s2 : e->maxd_s1;
  if (D > maxd) return LC_ERR_HEADDIM;  // the reference wrapper's `default:` branch
  return lc_attn_fwd_f16(Q, K, V, O, B, H, N, D, e->vt, e->family, e->acc_f32, st2, stream);
}

// ------------------------------------------------------------------------------------------------
// measurement helpers: every HIP return code is checked (a faulting kernel must not turn into a plausible ms value)
struct LcTimer {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipStream_t st = nullptr;
};

int lc_timer_start(void* stream, void** timer) {
  if (!timer) return LC_ERR_ARG;
  *timer = nullptr;
  LcTimer* t = new (std::nothrow) LcTimer();
  if (!t) return LC_ERR_LAUNCH;
  t->st = static_cast<hipStream_t>(stream);
  if (hipEventCreate(&t->e0) != hipSuccess) { delete t; return LC_ERR_LAUNCH; }
  if (hipEventCreate(&t->e1) != hipSuccess) { (void)hipEventDestroy(t->e0); delete t; return LC_ERR_LAUNCH; }
  if (hipEventRecord(t->e0, t->st) != hipSuccess) {
    (void)hipEventDestroy(t->e0); (void)hipEventDestroy(t->e1); delete t;
    return LC_ERR_LAUNCH;
  }
  *timer = t;
  return LC_OK;
}

int lc_timer_stop(void* timer, float* elapsed_ms) {
  LcTimer* t = static_cast<LcTimer*>(timer);
  if (!t) return LC_ERR_ARG;
  int rc = LC_OK;
  float ms = 0.f;
  if (hipEventRecord(t->e1, t->st) != hipSuccess) rc = LC_ERR_LAUNCH;
  if (rc == LC_OK && hipEventSynchronize(t->e1) != hipSuccess) rc = LC_ERR_LAUNCH;   // execution faults surface here
  if (rc == LC_OK && hipEventElapsedTime(&ms, t->e0, t->e1) != hipSuccess) rc = LC_ERR_LAUNCH;
  (void)hipEventDestroy(t->e0);
  (void)hipEventDestroy(t->e1);
  delete t;
  if (elapsed_ms) *elapsed_ms = ms;
  return (rc == LC_OK && !elapsed_ms) ? LC_ERR_ARG : rc;
}

int lc_hgemm_time(const void* A, const void* B, void* C, int M, int N, int K, int layout, int variant,
                  int stages, int swizzle_stride, int warmup, int iters, void* stream,
                  float* ms_per_launch) {
  if (!ms_per_launch || iters <= 0 || warmup < 0) return LC_ERR_ARG;
  for (int i = 0; i < warmup; ++i)
    if (int rc = lc_hgemm_f16(A, B, C, M, N, K, layout, variant, stages, swizzle_stride, stream)) return rc;
  void* t = nullptr;
  if (int rc = lc_timer_start(stream, &t)) return rc;
  int rc = LC_OK;
  for (int i = 0; i < iters && rc == LC_OK; ++i)
    rc = lc_hgemm_f16(A, B, C, M, N, K, layout, variant, stages, swizzle_stride, stream);
  float ms = 0.f;
  const int rc2 = lc_timer_stop(t, &ms);
  *ms_per_launch = ms / iters;
  return rc != LC_OK ? rc : rc2;
}

int lc_attn_time(const void* Q, const void* K, const void* V, void* O, int B, int H, int N, int D,
                 int v_transposed, int family, int stages, int warmup, int iters, void* stream,
                 float* ms_per_launch) {
  if (!ms_per_launch || iters <= 0 || warmup < 0) return LC_ERR_ARG;
  for (int i = 0; i < warmup; ++i)
    if (int rc = lc_attn_fwd_f16(Q, K, V, O, B, H, N, D, v_transposed, family, 0, stages, stream)) return rc;
  void* t = nullptr;
  if (int rc = lc_timer_start(stream, &t)) return rc;
  int rc = LC_OK;
  for (int i = 0; i < iters && rc == LC_OK; ++i)
    rc = lc_attn_fwd_f16(Q, K, V, O, B, H, N, D, v_transposed, family, 0, stages, stream);
  float ms = 0.f;
  const int rc2 = lc_timer_stop(t, &ms);
  *ms_per_launch = ms / iters;
  return rc != LC_OK ? rc : rc2;
}

// One-time calibration of the split-KV cost model on the CURRENT device (round 6): times, on zero-filled scratch tensors, the merged-phase
// kernel on two one-round grids that differ only in the number of KV tiles (tau_D = the difference per tile, D = 128 and 64) and two split
// launches of one shape (S = 2: one round of T / 2 tiles; S = 4: two rounds of T / 4) whose excess over the walk gives the fixed cost of the
// combine and the rate at which partials are written and read back.  About 60 launches of 20 ... 60 us + 70 MiB of scratch, freed before
// returning.  Values outside [0.4, 2.5] x the built-in constants are REFUSED (a busy or throttled GPU): the built-in constants then stay.
// out4 (optional): tau128, tau64, x0 (us), bytes per us — of what the rule will use from now on.  Returns LC_OK when the measured values
// were adopted, LC_ERR_DEVICE without a gfx950 device, LC_ERR_LAUNCH on a HIP error, LC_ERR_ARG when they were refused.
int lc_tune_calibrate(void* stream, float* out4) {
  int ncu = 0;
  if (int rc = lc_device_check(&ncu)) return rc;
  if (int rc = launch_guard()) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (stream_is_capturing(st)) return LC_ERR_ARG;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return LC_ERR_DEVICE;
  const int bh1 = ncu / 4 > 0 ? ncu / 4 : 1;                       // N = 1024: 4 blocks per head -> one round
  const size_t elems = (size_t)bh1 * 1024 * 128;                    // the largest tensor any step uses
  half_t* buf = nullptr;
  {
    RelaxedCaptureMode relaxed;
    if (hipMalloc(&buf, 4 * elems * sizeof(half_t)) != hipSuccess || !buf) {
      (void)hipGetLastError();
      return LC_ERR_LAUNCH;
    }
  }
  int rc = LC_OK;
  if (hipMemsetAsync(buf, 0, 4 * elems * sizeof(half_t), st) != hipSuccess) rc = LC_ERR_LAUNCH;
  half_t *q = buf, *k = buf + elems, *v = buf + 2 * elems, *o = buf + 3 * elems;
  auto time_us = [&](int D, int bh, int N, int walk, int ns, double* us) -> int {
    auto once = [&]() -> int {
      return D == 128 ? launch_attn_w4u_d128(q, k, v, o, 1, bh, N, walk, ns, st) : launch_attn_w4u_d64(q, k, v, o, 1, bh, N, walk, ns, st);
    };
    for (int i = 0; i < 3; ++i)
      if (int r = once()) return r;
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {   // best of three bursts of four: a burst shares its launch gaps, the minimum sheds a preempted one
      void* t = nullptr;
      if (int r = lc_timer_start(st, &t)) return r;
      int r2 = LC_OK;
      for (int i = 0; i < 4 && r2 == LC_OK; ++i) r2 = once();
      float ms = 0.f;
      const int r3 = lc_timer_stop(t, &ms);
      if (r2 != LC_OK) return r2;
      if (r3 != LC_OK) return r3;
      if (ms * 250.0 < best) best = ms * 250.0;   // us per launch
    }
    *us = best;
    return LC_OK;
  };
  double tau[2] = {0, 0}, x0 = 0, bpu = 0;
  for (int di = 0; di < 2 && rc == LC_OK; ++di) {
    const int D = di == 0 ? 128 : 64;
    double t16 = 0, t32 = 0;
    rc = time_us(D, bh1, 1024, 0, 1, &t16);                                  // ncu blocks x 16 tiles
    if (rc == LC_OK) rc = time_us(D, bh1 / 2 > 0 ? bh1 / 2 : 1, 2048, 0, 1, &t32);   // ncu blocks x 32 tiles
    tau[di] = (t32 - t16) / 16.0;
  }
  if (rc == LC_OK) {
    // (1, ncu / 16, 2048, 128): g = ncu / 2 blocks of T = 32 tiles; S = 2 -> one round of 16 tiles, S = 4 -> two rounds of 8
    const int bh = ncu / 16 > 0 ? ncu / 16 : 1;
    const double part = 4.0 * bh * 2048 * 128;
    double t2 = 0, t4 = 0;
    rc = time_us(128, bh, 2048, 3, 2, &t2);
    if (rc == LC_OK) rc = time_us(128, bh, 2048, 3, 4, &t4);
    const double e2 = t2 - 16.0 * tau[0], e4 = t4 - 2 * 8.0 * tau[0];       // = x0 + S part / bw
    const double per = (e4 - e2) / 2.0;                                       // part / bw
    bpu = per > 0 ? part / per : 0;
    x0 = e2 - 2.0 * per;
  }
  (void)hipStreamSynchronize(st);
  {
    RelaxedCaptureMode relaxed;
    (void)hipFree(buf);
  }
  if (rc != LC_OK) return rc;
  auto sane = [](double v, double ref) { return v >= 0.4 * ref && v <= 2.5 * ref; };
  const bool ok = sane(tau[0], 1.35) && sane(tau[1], 0.85) && sane(x0, kSplitFixedUs) && sane(bpu, kSplitBytesPerUs);
  AttnCalib& c = g_attn_calib[dev];
  if (ok) {
    c.valid.store(0, std::memory_order_release);
    c.tau128 = (float)tau[0];
    c.tau64 = (float)tau[1];
    c.x0 = (float)x0;
    c.bytes_per_us = (float)bpu;
    c.valid.store(1, std::memory_order_release);
  }
  if (out4) {
    out4[0] = (float)tau[0];
    out4[1] = (float)tau[1];
    out4[2] = (float)x0;
    out4[3] = (float)bpu;
  }
  return ok ? LC_OK : LC_ERR_ARG;
}

__global__ void lc_clock_probe_kernel(unsigned long long* out) {
  if (threadIdx.x == 0) {
    out[0] = __builtin_readcyclecounter();        // s_memtime: shader cycles
    out[1] = __builtin_amdgcn_s_memrealtime();    // constant 100 MHz
  }
}

int lc_clock_probe(void* out_u64x2, void* stream) {
  if (!out_u64x2) return LC_ERR_ARG;
  hipLaunchKernelGGL(lc_clock_probe_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream),
                     static_cast<unsigned long long*>(out_u64x2));
  return check_launch();
}

}  // extern "C"

extern "C" int lc_attn_slowpath_stats(unsigned* out4, int reset) {
  if (int rc = lc::diag_attn_slowpath_g(out4, reset)) return rc;
  if (int rc = lc::diag_attn_slowpath_u_d128(out4, reset)) return rc;
  if (int rc = lc::diag_attn_slowpath_u_d128t(out4, reset)) return rc;
  if (int rc = lc::diag_attn_slowpath_u_d64(out4, reset)) return rc;
  return lc::diag_attn_slowpath_u_d64t(out4, reset);
}

