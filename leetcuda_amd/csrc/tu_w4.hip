// tu_w4.hip — translation unit of the 4-wave HGEMM kernel (hgemm_w4.hip) — see lc_launch.h
#include "lc_launch.h"
#include "hgemm_w4y.hip"

namespace lc {
namespace {
template <bool B_KN, bool BUF, int DG>
int launch_w4_one(const half_t* A, const half_t* B, half_t* C, int M, int N, int K, int tiles_m, int tiles_n, int pw,
                  hipStream_t st) {
  auto kern = hgemm_w4b_kernel<B_KN, BUF, DG>;
  if (int rc = set_dyn_lds(kern, W4B_LDS)) return rc;
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), W4B_LDS, st, A, B, C, M, N, K, tiles_m, tiles_n, pw);
  return check_launch();
}

// nblk_arg: blocks to launch (< tiles_m * tiles_n when the ragged last wave goes to the 128-tile kernel), <= 0 = all; an explicit
// argument of every launcher below — no per-call state lives at file scope, so concurrent host threads cannot see each other's grid
template <bool B_KN, int Y>   // Y: -1 = hgemm_w4x_kernel, 0.. = hgemm_w4y_kernel<.., Y>
int launch_w4x_one(const half_t* A, const half_t* B, half_t* C, int M, int N, int K, int tiles_m, int tiles_n, int pw,
                   int nblk_arg, hipStream_t st) {
  auto kern = [] {
    if constexpr (Y < 0) return hgemm_w4x_kernel<B_KN>;
    else return hgemm_w4y_kernel<B_KN, Y>;
  }();
  if (int rc = set_dyn_lds(kern, W4B_LDS)) return rc;
  const int nblk = nblk_arg > 0 ? nblk_arg : tiles_m * tiles_n;
  if constexpr (Y < 0) {
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), W4B_LDS, st, A, B, C, M, N, K, tiles_m, tiles_n, pw);
  } else {
    // lc_tune_set "hgemm_persist" = 1: one workgroup per CU walks the block ids (hgemm_w4y.hip); only when every workgroup gets the
    // same number of tiles (a ragged walk would leave CUs idle for a whole tile)
    const int ncu = device_cu_count();
    const bool persist = g_tune_hgemm_persist != 0 && nblk > ncu && nblk % ncu == 0;
    hipLaunchKernelGGL(kern, dim3(persist ? ncu : nblk), dim3(256), W4B_LDS, st, A, B, C, M, N, K, tiles_m, tiles_n, pw,
                       stagger_arg(K / BK), persist ? nblk : 0);
  }
  return check_launch();
}

template <bool B_KN>
int launch_w4_t(const half_t* A, const half_t* B, half_t* C, int M, int N, int K, int variant, int tiles_m,
                int tiles_n, int pw, int nblk, hipStream_t st) {
  variant = w4_effective_variant(variant, B_KN, N, K);
#ifdef LC_DIAG
  if (variant == LC_HGEMM_MFMA256W4C && g_tune_hgemm_stamps)
    return launch_w4_one<B_KN, true, 1>(A, B, C, M, N, K, tiles_m, tiles_n, pw, st);
  if (variant == LC_HGEMM_MFMA256W4C && g_tune_w4_abl) {
    switch (g_tune_w4_abl) {   // bits: 2 no DMA, 4 no wait + barrier, 8 no fragment reads
      case 2: return launch_w4_one<B_KN, true, 2>(A, B, C, M, N, K, tiles_m, tiles_n, pw, st);
      case 4: return launch_w4_one<B_KN, true, 4>(A, B, C, M, N, K, tiles_m, tiles_n, pw, st);
      case 6: return launch_w4_one<B_KN, true, 6>(A, B, C, M, N, K, tiles_m, tiles_n, pw, st);
      case 8: return launch_w4_one<B_KN, true, 8>(A, B, C, M, N, K, tiles_m, tiles_n, pw, st);
      case 14: return launch_w4_one<B_KN, true, 14>(A, B, C, M, N, K, tiles_m, tiles_n, pw, st);
      default: return LC_ERR_ARG;
    }
  }
#endif
  if constexpr (!B_KN) {
    if (variant == LC_HGEMM_MFMA256W4X) return launch_w4x_one<B_KN, -1>(A, B, C, M, N, K, tiles_m, tiles_n, pw, nblk, st);
    if (variant == LC_HGEMM_MFMA256W4Y) {
      const int sched = g_tune_w4y_sched;   // read once per launch
      if (sched == 0) return launch_w4x_one<B_KN, 0>(A, B, C, M, N, K, tiles_m, tiles_n, pw, nblk, st);
      if (sched == 1) return launch_w4x_one<B_KN, 1>(A, B, C, M, N, K, tiles_m, tiles_n, pw, nblk, st);
#ifdef LC_DIAG
      if (sched == 3) return launch_w4x_one<B_KN, 3>(A, B, C, M, N, K, tiles_m, tiles_n, pw, nblk, st);
      if (sched == 4) return launch_w4x_one<B_KN, 4>(A, B, C, M, N, K, tiles_m, tiles_n, pw, nblk, st);
      if (sched == 5) return launch_w4x_one<B_KN, 5>(A, B, C, M, N, K, tiles_m, tiles_n, pw, nblk, st);
#endif
      return launch_w4x_one<B_KN, 2>(A, B, C, M, N, K, tiles_m, tiles_n, pw, nblk, st);
    }
  } else {
    if (variant == LC_HGEMM_MFMA256W4Y) return launch_w4x_one<B_KN, 1>(A, B, C, M, N, K, tiles_m, tiles_n, pw, nblk, st);   // one NN schedule
  }
  if (variant == LC_HGEMM_MFMA256W4C || variant == LC_HGEMM_MFMA256W4X || variant == LC_HGEMM_MFMA256W4Y)
    return launch_w4_one<B_KN, true, 0>(A, B, C, M, N, K, tiles_m, tiles_n, pw, st);
  return launch_w4_one<B_KN, false, 0>(A, B, C, M, N, K, tiles_m, tiles_n, pw, st);
}
}  // namespace

// buffer-descriptor DMA addresses are 32-bit offsets from the wave's first row: fall back to the 64-bit global form
// when an offset could reach 2 GiB (NN: K tiles step through the whole of B)
int w4_effective_variant(int variant, bool b_kn, int N, int K) {
  if (variant == LC_HGEMM_MFMA256W4X && b_kn) variant = LC_HGEMM_MFMA256W4C;   // the compiler-scheduled 16x16x32 kernel is TN only
  if (variant == LC_HGEMM_MFMA256W4C || variant == LC_HGEMM_MFMA256W4X ||
      variant == LC_HGEMM_MFMA256W4Y) {
    // (K-contiguous operands: a wave's pieces reach 64 rows past its base, 232 with hgemm_w4y's 32-row piece stride)
    const size_t rows_off = (size_t)K * 2 * (variant == LC_HGEMM_MFMA256W4Y ? 260 : 130);
    const size_t max_off = b_kn ? (size_t)K * N * 2 + (size_t)N * 64 : rows_off;
    if (max_off >= ((size_t)1 << 31) || rows_off >= ((size_t)1 << 31)) return LC_HGEMM_MFMA256W4B;
  }
  return variant;
}

int launch_w4_family(const half_t* A, const half_t* B, half_t* C, int M, int N, int K, int variant, bool b_kn,
                     int tiles_m, int tiles_n, int panel_w, int nblk, hipStream_t st) {
  const int nb = (nblk > 0 && nblk < tiles_m * tiles_n && w4_effective_variant(variant, b_kn, N, K) == LC_HGEMM_MFMA256W4Y) ? nblk : -1;
  return b_kn ? launch_w4_t<true>(A, B, C, M, N, K, variant, tiles_m, tiles_n, panel_w, nb, st)
              : launch_w4_t<false>(A, B, C, M, N, K, variant, tiles_m, tiles_n, panel_w, nb, st);
}
}  // namespace lc
