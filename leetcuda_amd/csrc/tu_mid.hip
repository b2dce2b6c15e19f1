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

// Tiles over the right strip (all rows, columns Ni .. N) and the bottom strip (rows Mi .. M, columns 0 .. Ni) of C, tiles that reach beyond M / N
// clamped and predicated (hgemm_mid_edge_kernel); Mi = Ni = 0: the whole of a ragged problem.  K % 32 == 0, K >= 64, N % 8 == 0, Ni a multiple of
// the tile width.  Tiles: 128 x 128 with 2 / 3 ring slots; with 3 slots also 64 x 128, 192 x 128 (NN), 128 x 192 and 192 x 192 (TN).
namespace {
template <bool B_KN, int TMW, int TNW, int NS>
int launch_mid_edge_one(const half_t* A, const half_t* B, half_t* C, int M, int N, int K, int Mi, int Ni, hipStream_t st) {
  using G = Mid<TMW, TNW, NS>;
  if (Ni % G::TN != 0) return LC_ERR_SHAPE;
  const long nrc = (N - Ni + G::TN - 1) / G::TN, nright = nrc * ((M + G::TM - 1) / G::TM), nbottom = (long)((M - Mi + G::TM - 1) / G::TM) * (Ni / G::TN);
  if (nright + nbottom <= 0) return LC_OK;
  if (nright + nbottom > INT_MAX) return LC_ERR_SHAPE;
  auto kern = hgemm_mid_edge_kernel<B_KN, TMW, TNW, NS>;
  if (int rc = set_dyn_lds(kern, G::LDS)) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)(nright + nbottom)), dim3(256), G::LDS, st, A, B, C, M, N, K, Mi, Ni, (int)nright, (int)(nrc > 0 ? nrc : 1));
  return check_launch();
}
}  // namespace
int launch_hgemm_mid_edge(const half_t* A, const half_t* B, half_t* C, int M, int N, int K, bool b_kn, int tmw, int tnw, int ns, int Mi, int Ni, hipStream_t st) {
  if (N % 8 != 0 || K % 32 != 0 || K < BK || K >= (1 << 22) || N >= (1 << 22) || Mi < 0 || Ni < 0 || Mi > M || Ni > N) return LC_ERR_SHAPE;
  const int code = 100 * tmw + 10 * tnw + ns;
  if (b_kn) {
    switch (code) {
      case 222: return launch_mid_edge_one<true, 2, 2, 2>(A, B, C, M, N, K, Mi, Ni, st);
      case 223: return launch_mid_edge_one<true, 2, 2, 3>(A, B, C, M, N, K, Mi, Ni, st);
      case 123: return launch_mid_edge_one<true, 1, 2, 3>(A, B, C, M, N, K, Mi, Ni, st);
      case 323: return launch_mid_edge_one<true, 3, 2, 3>(A, B, C, M, N, K, Mi, Ni, st);
      default: return LC_ERR_ARG;
    }
  }
  switch (code) {
    case 222: return launch_mid_edge_one<false, 2, 2, 2>(A, B, C, M, N, K, Mi, Ni, st);
    case 223: return launch_mid_edge_one<false, 2, 2, 3>(A, B, C, M, N, K, Mi, Ni, st);
    case 123: return launch_mid_edge_one<false, 1, 2, 3>(A, B, C, M, N, K, Mi, Ni, st);
    case 233: return launch_mid_edge_one<false, 2, 3, 3>(A, B, C, M, N, K, Mi, Ni, st);
    case 333: return launch_mid_edge_one<false, 3, 3, 3>(A, B, C, M, N, K, Mi, Ni, st);
    default: return LC_ERR_ARG;
  }
}

// Split-K of a whole ragged problem on hgemm_mid_edge_sk_kernel (64 / 128 x 128 tiles, three ring slots): part = ks x Mp x Np floats, Mp x Np = the
// tile grid's extent (launch_hgemm_mid_edge_sk_floats), then the reduce.  ks >= 2, at least two K tiles per range.
namespace {
template <bool B_KN, int TMW>
int launch_mid_edge_sk(const half_t* A, const half_t* B, half_t* C, int M, int N, int K, float* part, int ks, hipStream_t st) {
  using G = Mid<TMW, 2, 3>;
  const long tm = (M + G::TM - 1) / G::TM, tn = (N + G::TN - 1) / G::TN;
  if (tm * tn * ks > INT_MAX) return LC_ERR_SHAPE;
  auto kern = hgemm_mid_edge_sk_kernel<B_KN, TMW, 3>;
  if (int rc = set_dyn_lds(kern, G::LDS)) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)(tm * tn * ks)), dim3(256), G::LDS, st, A, B, M, N, K, (int)(tm * tn), (int)tn, part, ks);
  if (int rc = check_launch()) return rc;
  const size_t chunks = (size_t)M * (N / 8);
  hipLaunchKernelGGL(hgemm_mid_reduce_edge_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, st, (const float*)part, C, M, N, (size_t)tm * G::TM,
                     (size_t)tn * G::TN, ks);
  return check_launch();
}
}  // namespace
size_t launch_hgemm_mid_edge_sk_floats(int M, int N, int tmw, int ks) {
  const size_t tm = (size_t)(M + 64 * tmw - 1) / (64 * tmw), tn = (size_t)(N + 127) / 128;
  return (size_t)ks * tm * (64 * tmw) * tn * 128;
}
int launch_hgemm_mid_edge_sk(const half_t* A, const half_t* B, half_t* C, int M, int N, int K, bool b_kn, int tmw, int ks, float* part, hipStream_t st) {
  if (N % 8 != 0 || K % 32 != 0 || K < BK || K >= (1 << 22) || N >= (1 << 22)) return LC_ERR_SHAPE;
  if (!part || (tmw != 1 && tmw != 2) || ks < 2 || ks > 16 || K / BK < 2 * ks) return LC_ERR_ARG;
  if (b_kn) return tmw == 1 ? launch_mid_edge_sk<true, 1>(A, B, C, M, N, K, part, ks, st) : launch_mid_edge_sk<true, 2>(A, B, C, M, N, K, part, ks, st);
  return tmw == 1 ? launch_mid_edge_sk<false, 1>(A, B, C, M, N, K, part, ks, st) : launch_mid_edge_sk<false, 2>(A, B, C, M, N, K, part, ks, st);
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
