// tu_mid.hip — translation unit of the mid-size HGEMM kernel (hgemm_mid.hip) — see lc_launch.h
#include <climits>
#include "lc_launch.h"
#include "hgemm_mid.hip"

namespace lc {
namespace {
template <bool B_KN, int TMW, int TNW, int NS>
int launch_mid_one(const half_t* A, const half_t* B, half_t* C, int M, int N, int K, int pw, hipStream_t st) {
  using G = Mid<TMW, TNW, NS>;
  auto kern = hgemm_mid_kernel<B_KN, TMW, TNW, NS>;
  if (int rc = set_dyn_lds(kern, G::LDS)) return rc;
  const int tiles_m = M / G::TM, tiles_n = N / G::TN;
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), G::LDS, st, A, B, C, M, N, K, tiles_m, tiles_n, pw, -1);
  return check_launch();
}
template <bool B_KN, int TMW, int NS>
int launch_mid_rem_one(const half_t* A, const half_t* B, half_t* C, int M, int N, int K, int tiles_m256, int tiles_n256, int pw256, int rem_base,
                       int rem_tiles, hipStream_t st) {
  using G = Mid<TMW, 2, NS>;
  auto kern = hgemm_mid_kernel<B_KN, TMW, 2, NS>;
  if (int rc = set_dyn_lds(kern, G::LDS)) return rc;
  hipLaunchKernelGGL(kern, dim3(rem_tiles * (256 / G::TM) * 2), dim3(256), G::LDS, st, A, B, C, M, N, K, tiles_m256, tiles_n256, pw256, rem_base);
  return check_launch();
}
template <bool B_KN>
int launch_mid_rem(const half_t* A, const half_t* B, half_t* C, int M, int N, int K, int tmw, int ns, int tiles_m256, int tiles_n256, int pw256,
                   int rem_base, int rem_tiles, hipStream_t st) {
  if (tmw == 1)
    return ns == 3 ? launch_mid_rem_one<B_KN, 1, 3>(A, B, C, M, N, K, tiles_m256, tiles_n256, pw256, rem_base, rem_tiles, st)
                   : launch_mid_rem_one<B_KN, 1, 2>(A, B, C, M, N, K, tiles_m256, tiles_n256, pw256, rem_base, rem_tiles, st);
  return ns == 3 ? launch_mid_rem_one<B_KN, 2, 3>(A, B, C, M, N, K, tiles_m256, tiles_n256, pw256, rem_base, rem_tiles, st)
                 : launch_mid_rem_one<B_KN, 2, 2>(A, B, C, M, N, K, tiles_m256, tiles_n256, pw256, rem_base, rem_tiles, st);
}
// split-K: ks copies of the tile grid (three ring slots: the grids it serves are one round), fp32 partials in `part`, then the reduce
template <bool B_KN, int TMW>
int launch_mid_sk(const half_t* A, const half_t* B, half_t* C, int M, int N, int K, int pw, float* part, int ks, hipStream_t st) {
  using G = Mid<TMW, 2, 3>;
  auto kern = hgemm_mid_sk_kernel<B_KN, TMW, 3>;
  if (int rc = set_dyn_lds(kern, G::LDS)) return rc;
  const int tiles_m = M / G::TM, tiles_n = N / G::TN;
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n * ks), dim3(256), G::LDS, st, A, B, M, N, K, tiles_m, tiles_n, pw, part, ks);
  if (int rc = check_launch()) return rc;
  const size_t mn = (size_t)M * N;
  hipLaunchKernelGGL(hgemm_mid_reduce_kernel, dim3((unsigned)((mn / 8 + 255) / 256)), dim3(256), 0, st, (const float*)part, C, mn, ks);
  return check_launch();
}
template <bool B_KN, int TMW, int TNW>
int launch_mid_ns(const half_t* A, const half_t* B, half_t* C, int M, int N, int K, int ns, int pw, hipStream_t st) {
  if (ns == 2) return launch_mid_one<B_KN, TMW, TNW, 2>(A, B, C, M, N, K, pw, st);
  if (ns == 3) return launch_mid_one<B_KN, TMW, TNW, 3>(A, B, C, M, N, K, pw, st);
  return LC_ERR_ARG;
}
template <bool B_KN, int TNW>
int launch_mid_tm(const half_t* A, const half_t* B, half_t* C, int M, int N, int K, int tmw, int ns, int pw, hipStream_t st) {
  if (tmw == 1) return launch_mid_ns<B_KN, 1, TNW>(A, B, C, M, N, K, ns, pw, st);
  if (tmw == 3) return launch_mid_ns<B_KN, 3, TNW>(A, B, C, M, N, K, ns, pw, st);   // 192 x 128 (the NN counterpart of 128 x 192: NN has no 192-wide transpose image), 192 x 192 (TN)
  return launch_mid_ns<B_KN, 2, TNW>(A, B, C, M, N, K, ns, pw, st);
}
}  // namespace

// tmw: 1 / 2 / 3 = 64 / 128 / 192 tile rows; tnw: 2 / 3 = 128 / 192 tile columns (NN: 2); ns: ring slots 2 / 3; pw: block -> tile map (panel_tiles, lc_abi.hip)
// part / ks: split-K (ks >= 2: tmw 1 / 2, tnw 2, at least two K tiles per range; part = ks x M x N floats)
int launch_hgemm_mid(const half_t* A, const half_t* B, half_t* C, int M, int N, int K, bool b_kn, int tmw, int tnw, int ns, int pw,
                     hipStream_t st, float* part, int ks) {
  if (tmw < 1 || tmw > 3 || tnw < 2 || tnw > 3 || (b_kn && tnw != 2)) return LC_ERR_ARG;
  if (M % (64 * tmw) != 0 || N % (64 * tnw) != 0 || K % 32 != 0 || K < BK || K >= (1 << 22) || N >= (1 << 22)) return LC_ERR_SHAPE;
  if (ks > 1) {
    if (!part || tmw > 2 || tnw != 2 || ks > 16 || K / BK < 2 * ks || (size_t)(M / (64 * tmw)) * (N / 128) * ks > (size_t)INT_MAX) return LC_ERR_ARG;
    if (b_kn) return tmw == 1 ? launch_mid_sk<true, 1>(A, B, C, M, N, K, pw, part, ks, st) : launch_mid_sk<true, 2>(A, B, C, M, N, K, pw, part, ks, st);
    return tmw == 1 ? launch_mid_sk<false, 1>(A, B, C, M, N, K, pw, part, ks, st) : launch_mid_sk<false, 2>(A, B, C, M, N, K, pw, part, ks, st);
  }
  if (b_kn) return launch_mid_tm<true, 2>(A, B, C, M, N, K, tmw, ns, pw, st);
  if (tnw == 2) return launch_mid_tm<false, 2>(A, B, C, M, N, K, tmw, ns, pw, st);
  return launch_mid_tm<false, 3>(A, B, C, M, N, K, tmw, ns, pw, st);
}

// The ragged last round of hgemm_w4y_kernel's 256 x 256 grid as 128 x 128 quadrants (tmw = 2) or 64 x 128 eighths (tmw = 1) on this kernel
// (round 6; until then on hgemm_mfma128_kernel with a workspace split-K): rem_tiles 256-tiles from raster id rem_base on, tiles_m256 /
// tiles_n256 / pw256 = that grid's dimensions and block map.  ns: ring slots (3 when the blocks fit one round of the CUs, else 2).
int launch_hgemm_mid_rem(const half_t* A, const half_t* B, half_t* C, int M, int N, int K, bool b_kn, int tmw, int ns, int tiles_m256,
                         int tiles_n256, int pw256, int rem_base, int rem_tiles, hipStream_t st) {
  if (K % 32 != 0 || K < BK || K >= (1 << 22) || N >= (1 << 22) || rem_base < 0 || rem_tiles <= 0) return LC_ERR_SHAPE;
  if ((tmw != 1 && tmw != 2) || (ns != 2 && ns != 3)) return LC_ERR_ARG;
  return b_kn ? launch_mid_rem<true>(A, B, C, M, N, K, tmw, ns, tiles_m256, tiles_n256, pw256, rem_base, rem_tiles, st)
              : launch_mid_rem<false>(A, B, C, M, N, K, tmw, ns, tiles_m256, tiles_n256, pw256, rem_base, rem_tiles, st);
}
}  // namespace lc
