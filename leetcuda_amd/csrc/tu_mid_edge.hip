// tu_mid_edge.hip — translation unit of the mid-size HGEMM kernel's EDGE instantiations (hgemm_mid.hip: tiles that reach beyond M / N; LC_HGEMM_RAGGED,
// lc_abi.hip launch_ragged) — a unit of its own so that the build's longest compile is split in two; see lc_launch.h
#include <climits>
#include "lc_launch.h"
#include "hgemm_mid.hip"

namespace lc {
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
}  // namespace lc
