// lc_diag.hip — liblc_diag.so: the hardware layout / issue-overlap probes declared in include/lc_diag.h.
// Test and measurement infrastructure only; the drop-in library (libleetcuda_amd.so) does not contain or need it.
#include "../../../include/lc_diag.h"

#include <hip/hip_runtime.h>

#include "probe_kernels.hip"

using namespace lc;

namespace {
constexpr int OK = 0, ERR_ARG = -1, ERR_LAUNCH = -4;   // lc_status values (include/lc_abi.h)
inline int check_launch() { return hipGetLastError() == hipSuccess ? OK : ERR_LAUNCH; }

template <int F, int K>
int probe_coissue_mode(int mode, unsigned long long* out, hipStream_t st) {
  if (mode == 0) hipLaunchKernelGGL((probe_coissue_kernel<F, K, 0>), dim3(1), dim3(256), 0, st, out, 1.0f);
  else if (mode == 1) return ERR_ARG;   // (cross-wave mode removed: its per-instruction branches dominated the result)
  else if (mode == 3) hipLaunchKernelGGL((probe_coissue_kernel<F, K, 3>), dim3(1), dim3(256), 0, st, out, 1.0f);
  else hipLaunchKernelGGL((probe_coissue_kernel<F, K, 2>), dim3(1), dim3(256), 0, st, out, 1.0f);
  return check_launch();
}
template <int F>
int probe_coissue_k(int k, int mode, unsigned long long* out, hipStream_t st) {
  switch (k) {
    case 1: return probe_coissue_mode<F, 1>(mode, out, st);
    case 2: return probe_coissue_mode<F, 2>(mode, out, st);
    case 4: return probe_coissue_mode<F, 4>(mode, out, st);
    case 8: return probe_coissue_mode<F, 8>(mode, out, st);
    default: return ERR_ARG;
  }
}
template <int KIND, int QUEUED>
int probe_war_delay(int delay, const half_t* a, const half_t* b, float* d, hipStream_t st) {
#define LC_WAR_CASE(D) case D: hipLaunchKernelGGL((probe_mfma_war_kernel<D, KIND, QUEUED>), dim3(1), dim3(64), 0, st, a, b, d); break;
  switch (delay) {
    LC_WAR_CASE(0) LC_WAR_CASE(1) LC_WAR_CASE(2) LC_WAR_CASE(3) LC_WAR_CASE(4) LC_WAR_CASE(6) LC_WAR_CASE(8)
    LC_WAR_CASE(11) LC_WAR_CASE(15)
    default: return ERR_ARG;
  }
#undef LC_WAR_CASE
  return check_launch();
}
}  // namespace

extern "C" {

int lc_probe_mfma16(const void* a, const void* b, float* d, void* stream) {
  if (!a || !b || !d) return ERR_ARG;
  hipLaunchKernelGGL(probe_mfma16_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream),
                     static_cast<const half_t*>(a), static_cast<const half_t*>(b), d);
  return check_launch();
}
int lc_probe_mfma32(const void* a, const void* b, float* d, void* stream) {
  if (!a || !b || !d) return ERR_ARG;
  hipLaunchKernelGGL(probe_mfma32_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream),
                     static_cast<const half_t*>(a), static_cast<const half_t*>(b), d);
  return check_launch();
}
int lc_probe_tr16(const void* src, void* dst, void* stream) {
  if (!src || !dst) return ERR_ARG;
  hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream),
                     static_cast<const uint16_t*>(src), static_cast<uint16_t*>(dst));
  return check_launch();
}
int lc_probe_coissue(int filler, int k, int mode, void* out_u64x16, void* stream) {
  if (!out_u64x16 || mode < 0 || mode > 3) return ERR_ARG;
  unsigned long long* out = static_cast<unsigned long long*>(out_u64x16);
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (filler) {
    case 0: return probe_coissue_mode<0, 1>(mode == 2 ? 0 : mode, out, st);
    case 1: return probe_coissue_k<1>(k, mode, out, st);
    case 2: return probe_coissue_k<2>(k, mode, out, st);
    case 3: return probe_coissue_k<3>(k, mode, out, st);
    case 4: return probe_coissue_k<4>(k, mode, out, st);
    case 5: return probe_coissue_k<5>(k, mode, out, st);
    case 6: return probe_coissue_k<6>(k, mode, out, st);
    case 7: return probe_coissue_k<7>(k, mode, out, st);
    case 8: return probe_coissue_k<8>(k, mode, out, st);
    default: return ERR_ARG;
  }
}
// form 0..4: register files of the MFMA operands (probe_mfma_form_kernel); out[wave] = cycles of 4096 MFMAs
int lc_probe_mfma_form(int form, void* out_u64x16, void* stream) {
  if (!out_u64x16 || form < 0 || form > 4) return ERR_ARG;
  unsigned long long* out = static_cast<unsigned long long*>(out_u64x16);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (form == 0) hipLaunchKernelGGL(probe_mfma_form_kernel<0>, dim3(1), dim3(256), 0, st, out, 1.0f);
  else if (form == 1) hipLaunchKernelGGL(probe_mfma_form_kernel<1>, dim3(1), dim3(256), 0, st, out, 1.0f);
  else if (form == 2) hipLaunchKernelGGL(probe_mfma_form_kernel<2>, dim3(1), dim3(256), 0, st, out, 1.0f);
  else if (form == 3) hipLaunchKernelGGL(probe_mfma_form_kernel<3>, dim3(1), dim3(256), 0, st, out, 1.0f);
  else hipLaunchKernelGGL(probe_mfma_form_kernel<4>, dim3(1), dim3(256), 0, st, out, 1.0f);
  return check_launch();
}
// waves: 4 or 8 per workgroup (one workgroup on one CU); mix: 0..8 (probe_attn_mix_kernel)
int lc_probe_attn_mix(int waves, int mix, void* out_u64x16, void* stream) {
  if (!out_u64x16 || (waves != 4 && waves != 8) || mix < 0 || mix > 8) return ERR_ARG;
  unsigned long long* out = static_cast<unsigned long long*>(out_u64x16);
  hipStream_t st = static_cast<hipStream_t>(stream);
#define LC_MIX(W, M) hipLaunchKernelGGL((probe_attn_mix_kernel<W, M>), dim3(1), dim3(W * 64), 0, st, out, 1.0f)
  if (waves == 4) {
    if (mix == 0) LC_MIX(4, 0); else if (mix == 1) LC_MIX(4, 1); else if (mix == 2) LC_MIX(4, 2); else if (mix == 3) LC_MIX(4, 3); else if (mix == 4) LC_MIX(4, 4);
    else if (mix == 5) LC_MIX(4, 5); else if (mix == 6) LC_MIX(4, 6); else if (mix == 7) LC_MIX(4, 7); else LC_MIX(4, 8);
  } else {
    if (mix == 0) LC_MIX(8, 0); else if (mix == 1) LC_MIX(8, 1); else if (mix == 2) LC_MIX(8, 2); else if (mix == 3) LC_MIX(8, 3); else if (mix == 4) LC_MIX(8, 4);
    else if (mix == 5) LC_MIX(8, 5); else if (mix == 6) LC_MIX(8, 6); else if (mix == 7) LC_MIX(8, 7); else LC_MIX(8, 8);
  }
#undef LC_MIX
  return check_launch();
}
int lc_probe_mfma_war(int delay, int kind, int queued, const void* a, const void* b, float* d, void* stream) {
  if (!a || !b || !d) return ERR_ARG;
  const half_t* ah = static_cast<const half_t*>(a);
  const half_t* bh = static_cast<const half_t*>(b);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (kind == 0 && queued == 0) return probe_war_delay<0, 0>(delay, ah, bh, d, st);
  if (kind == 0 && queued == 1) return probe_war_delay<0, 1>(delay, ah, bh, d, st);
  if (kind == 1 && queued == 0) return probe_war_delay<1, 0>(delay, ah, bh, d, st);
  if (kind == 1 && queued == 1) return probe_war_delay<1, 1>(delay, ah, bh, d, st);
  if (kind == 2 && queued == 0) return probe_war_delay<2, 0>(delay, ah, bh, d, st);
  if (kind == 2 && queued == 1) return probe_war_delay<2, 1>(delay, ah, bh, d, st);
  if (kind == 2 && queued == 2) return probe_war_delay<2, 2>(delay, ah, bh, d, st);
  if (kind == 2 && queued == 4) return probe_war_delay<2, 4>(delay, ah, bh, d, st);
  if (kind == 0 && queued == 4) return probe_war_delay<0, 4>(delay, ah, bh, d, st);
  return ERR_ARG;
}

}  // extern "C"

int lc_diag_pollute(unsigned pattern, int what, void* stream) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(pollute_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 163840) != hipSuccess)
      return ERR_LAUNCH;
    attr_set = true;
  }
  // 8 workgroups per CU, one resident at a time (160 KiB of LDS, 512 registers): every SIMD's file and every LDS is visited
  hipLaunchKernelGGL(pollute_kernel, dim3(2048), dim3(256), 163840, static_cast<hipStream_t>(stream), pattern, what, nullptr);
  return check_launch();
}


// ---- ablated copies of the generated attention stream (attn_w4i_abl.hip, one translation unit per ablation)
extern "C" {
int lc_diag_attn_w4i_abl1(const void*, const void*, const void*, void*, int, int, int, int, void*);
int lc_diag_attn_w4i_abl2(const void*, const void*, const void*, void*, int, int, int, int, void*);
int lc_diag_attn_w4i_abl3(const void*, const void*, const void*, void*, int, int, int, int, void*);
int lc_diag_attn_w4i_abl4(const void*, const void*, const void*, void*, int, int, int, int, void*);
int lc_diag_attn_w4i_abl7(const void*, const void*, const void*, void*, int, int, int, int, void*);
int lc_diag_attn_w4i_abl8(const void*, const void*, const void*, void*, int, int, int, int, void*);
int lc_diag_attn_w4i_abl23(const void*, const void*, const void*, void*, int, int, int, int, void*);
int lc_diag_attn_w4i_abl55(const void*, const void*, const void*, void*, int, int, int, int, void*);

int lc_diag_attn_w4i(int abl, const void* Q, const void* K, const void* V, void* O, int B, int H, int N, int D, void* stream) {
  switch (abl) {
    case 1: return lc_diag_attn_w4i_abl1(Q, K, V, O, B, H, N, D, stream);
    case 2: return lc_diag_attn_w4i_abl2(Q, K, V, O, B, H, N, D, stream);
    case 3: return lc_diag_attn_w4i_abl3(Q, K, V, O, B, H, N, D, stream);
    case 4: return lc_diag_attn_w4i_abl4(Q, K, V, O, B, H, N, D, stream);
    case 7: return lc_diag_attn_w4i_abl7(Q, K, V, O, B, H, N, D, stream);
    case 8: return lc_diag_attn_w4i_abl8(Q, K, V, O, B, H, N, D, stream);
    case 23: return lc_diag_attn_w4i_abl23(Q, K, V, O, B, H, N, D, stream);
    case 55: return lc_diag_attn_w4i_abl55(Q, K, V, O, B, H, N, D, stream);
    default: return ERR_ARG;
  }
}
}
