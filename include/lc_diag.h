/*
 * lc_diag.h — diagnosis surface of the MI355X kernels.  NOT part of the drop-in boundary (include/lc_abi.h):
 * nothing here is needed to run the reference's entry points.
 *
 * 1. Hardware layout probes: tiny kernels that dump MFMA / LDS-transpose lane maps and issue-overlap timings.
 *    They live in their own library, leetcuda_amd/lib/liblc_diag.so (csrc/diag/lc_diag.hip), used by
 *    tests/test_gpu_probe.py and tools/{coissue,mfma_war}_probe.py only.
 * 2. Diagnosis keys of lc_tune_set(): compiled into libleetcuda_amd.so only when it is built with LC_DIAG=1
 *    (`LC_DIAG=1 python -m leetcuda_amd.build --force`); a production library rejects them with LC_ERR_ARG and
 *    lc_build_info() reports which kind a given .so is.  With an ablation key set, results are WRONG by design
 *    (work is skipped to price it) and the stamp keys overwrite the first bytes of A / Q with cycle counters.
 *      "attn_ablate"   attention ablation / stamp instantiations (tools/attn_ablate.py, tools/attn_*_stamps.py)
 *      "w4_abl"        4-wave HGEMM: 2 = no DMA after the prologue, 4 = no per-tile wait + barrier, 8 = no fragment
 *                      reads, 14 = MFMA issue only (tools/w4_ablate.py)
 *      "hgemm_stamps"  s_memtime stamps of one wave at the k-step boundaries (tools/hgemm_w4c_stamps.py)
 */
#ifndef LC_DIAG_H_
#define LC_DIAG_H_

#ifdef __cplusplus
extern "C" {
#endif

/* status codes as in lc_abi.h (0 = ok, -1 = bad argument, -4 = launch failed) */
int lc_probe_mfma16(const void* a16x32, const void* b16x32, float* d16x16, void* stream);
int lc_probe_mfma32(const void* a32x16, const void* b32x16, float* d32x32, void* stream);
int lc_probe_tr16(const void* src_64x4_u16, void* dst_64x4_u16, void* stream);
/* issue-overlap micro-benchmark (tools/coissue_probe.py): 256 x 4 x { 1 MFMA 32x32x16 f16, k fillers }; filler 0 none,
 * 1 v_fma_f32, 2 v_exp_f32, 3 v_pk_fma_f32, 4 v_cvt_pk_f16_f32, 5 ds_read_b128; mode 0 same wave, 2 fillers only;
 * out = 16 x u64 (s_memtime cycles per wave). */
int lc_probe_coissue(int filler, int k, int mode, void* out_u64x16, void* stream);
/* one workgroup of `waves` (4 or 8) wave64 on one CU, each running v_mfma_f32_16x16x32_f16 + the softmax share of an MFMA slot
 * (mix 0 none, 1 D = 128, 2 D = 64, 3 / 4 = 2 / 1 without the LDS read, 5 / 6 = 3 / 4 with exp2 as a packed-fp16 polynomial instead of v_exp_f32):
 * out[wave] = cycles of 2048 MFMAs (tools/attn_mix_probe.py) */
int lc_probe_attn_mix(int waves, int mix, void* out_u64x16, void* stream);
/* attn_fwd_w4u_kernel<128, false, 0 | 3> compiled with cycle stamps (csrc/diag/attn_w4u_stamps.hip): wave 0 of workgroup 0 records s_memtime at the
 * milestones of its block — out[0 .. 11] shader cycles (entry, requests issued, Q parked, first tiles landed, first S, tile 0, tiles 1 .. T - 2, last
 * tile, epilogue barrier, O staged, O stores issued / acknowledged), out[14] / out[15] s_memrealtime (100 MHz) at entry / exit.  nsplit == 1: whole
 * heads into O; nsplit >= 2: the split-KV walk into partials O [nsplit][B H][N][128] + lse [nsplit][B H][N] (tools/attn_w4u_stamps.py) */
int lc_diag_attn_w4u_stamps(const void* Q, const void* K, const void* V, void* O, void* lse, int B, int H, int N, int nsplit, void* out_u64x16,
                            void* stream);
/* v_mfma_f32_16x16x32_f16 with operands from VGPRs / AGPRs (form 0 all VGPR, 1 A/B AGPR + C/D VGPR, 2 A/B VGPR + C/D AGPR, 3 = 1 and 2
 * alternating, 4 all AGPR): out[wave] = cycles of 4096 MFMAs, one wave per SIMD (tools/attn_mix_probe.py) */
int lc_probe_mfma_form(int form, void* out_u64x16, void* stream);
/* does an in-flight 32x32x16 MFMA still read its A operand after issue? (tools/mfma_war_probe.py): the A registers are
 * overwritten by VALU `delay`+1 wait states after the MFMA (kind 0 v_mov, 1 v_exp_f32; queued: behind another MFMA). */
int lc_probe_mfma_war(int delay, int kind, int queued, const void* a32x16, const void* b32x16, float* d32x32,
                      void* stream);

/* leave `pattern` in every arch VGPR (what & 1), AGPR (what & 2) and LDS dword (what & 4) of the chip: a following kernel
 * that reads state it did not initialise inherits it (tools/attn_determinism.py) */
int lc_diag_pollute(unsigned pattern, int what, void* stream);

/* attn_fwd_w4i_kernel (D = 64 / 128, N % 256 == 0) with a class of instructions REMOVED from the generated phase statements
 * (tools/gen_attn_w4i.py --diag): abl bits 1 = no LDS-DMA, 2 = no LDS reads, 4 = no softmax VALU, 8 = no MFMA, 16 = no per-tile
 * wait + barrier, 32 = no guard decision; available: 1, 2, 3, 4, 7, 8, 23, 55.  Results are WRONG by design: timing only (tools/attn_w4i_ablate.py). */
int lc_diag_attn_w4i(int abl, const void* Q, const void* K, const void* V, void* O, int B, int H, int N, int D, void* stream);

/* hgemm_mid_kernel<B_KN, 4, 4, 2> (csrc/diag/mid256_probe.hip): the mid-size kernel's rotated hand-ordered loop on a 256 x 256 tile with two ring slots.
 * M, N % 256 == 0, K % 64 == 0; panel_w: tiles per N panel of the block swizzle.  0 on success. */
int lc_probe_mid256(const void* A, const void* B, void* C, int M, int N, int K, int b_kn, int panel_w, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LC_DIAG_H_ */
