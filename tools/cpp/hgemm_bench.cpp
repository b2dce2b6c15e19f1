// hgemm_bench.cpp — standalone C++ bench + error-check harness for the HGEMM path.
//
// What the reference ships as `main()` at the tail of its HGEMM .cu files plus kernels/hgemm/utils/utils.h
// (perf_gemm :20-58, gemm_error_check_nn / _tn :93-277, driver loop hgemm_mma_stage.cu:1963-2036, makefile:10-17):
// a torch-free binary that (1) checks the kernel against the vendor GEMM on seeded uniform{-1 .. 0.99}/0.01 inputs and
// (2) times M = N = K = 256, 512, ... with device events.  Same protocol here, on the C-ABI only
// (libleetcuda_amd.so + the HIP runtime; no PyTorch in the measurement loop), with two differences the round-1 verdict
// asked for: the inputs are SEEDED (the reference calls srand(time(0))) and the error check has a THRESHOLD and an exit
// code (the reference only prints the maximum).
//
//   hgemm_bench [--layout nn|tn] [--variant N] [--max-n 16384] [--check-n 5] [--outer 10] [--inner 1] [--warmup 1]
//               [--stride 2048] [--mnk M N K]
#include <hip/hip_runtime_api.h>

#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "lc_abi.h"

#define HIP_OK(x)                                                                          \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) {                                                                \
      fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(3);                                                                             \
    }                                                                                      \
  } while (0)
#define LC_OKAY(x)                                                                         \
  do {                                                                                     \
    int s_ = (x);                                                                          \
    if (s_ != LC_OK) {                                                                     \
      fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #x, lc_status_string(s_));  \
      exit(4);                                                                             \
    }                                                                                      \
  } while (0)

static uint16_t f2h(float f) {   // float -> IEEE half, round to nearest even (inputs here are small decimals)
  uint32_t x;
  memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  int32_t e = (int32_t)((x >> 23) & 0xff) - 127 + 15;
  uint32_t m = x & 0x7fffffu;
  if (e >= 31) return (uint16_t)(sign | 0x7c00u);
  if (e <= 0) {
    if (e < -10) return (uint16_t)sign;
    m |= 0x800000u;
    const int sh = 14 - e;
    uint32_t h = m >> sh;
    const uint32_t rem = m & ((1u << sh) - 1), half = 1u << (sh - 1);
    if (rem > half || (rem == half && (h & 1))) ++h;
    return (uint16_t)(sign | h);
  }
  uint32_t h = ((uint32_t)e << 10) | (m >> 13);
  const uint32_t rem = m & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) ++h;
  return (uint16_t)(sign | h);
}
static float h2f(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t e = (h >> 10) & 0x1f, m = h & 0x3ffu, x;
  if (e == 0) {
    if (m == 0) x = sign;
    else {
      e = 127 - 15 + 1;
      while (!(m & 0x400u)) { m <<= 1; --e; }
      x = sign | (e << 23) | ((m & 0x3ffu) << 13);
    }
  } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
  else x = sign | ((e - 15 + 127) << 23) | (m << 13);
  float f;
  memcpy(&f, &x, 4);
  return f;
}

struct Opts {
  int layout = LC_LAYOUT_NN, variant = LC_HGEMM_AUTO, max_n = 16384, check_n = 5, outer = 10, inner = 1, warmup = 1;
  int stride = 2048, M = 0, N = 0, K = 0;
};

// utils.h:238-241: (rand() % 200 - 100) * 0.01, here with a FIXED seed per (M,N,K)
static void fill(std::vector<uint16_t>& v, unsigned seed) {
  srand(seed);
  for (auto& x : v) x = f2h((float)(rand() % 200 - 100) * 0.01f);
}

struct DevBufs {
  void *a = nullptr, *b = nullptr, *c = nullptr, *cref = nullptr;
  DevBufs(size_t na, size_t nb, size_t nc) {
    HIP_OK(hipMalloc(&a, na * 2));
    HIP_OK(hipMalloc(&b, nb * 2));
    HIP_OK(hipMalloc(&c, nc * 2));
    HIP_OK(hipMalloc(&cref, nc * 2));
  }
  ~DevBufs() { (void)hipFree(a); (void)hipFree(b); (void)hipFree(c); (void)hipFree(cref); }
};

// gemm_error_check_nn / _tn: our kernel against the vendor GEMM (hipBLASLt behind the reference's cuBLAS entry points)
static bool error_check(const Opts& o, int M, int N, int K) {
  std::vector<uint16_t> ha((size_t)M * K), hb((size_t)K * N), hc((size_t)M * N), hr((size_t)M * N);
  fill(ha, 1u + (unsigned)M);
  fill(hb, 7u + (unsigned)N);   // TN: the same bytes read as the [N,K] storage of B
  DevBufs d(ha.size(), hb.size(), hc.size());
  HIP_OK(hipMemcpy(d.a, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d.b, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
  HIP_OK(hipMemset(d.c, 0xff, hc.size() * 2));
  LC_OKAY(lc_hgemm_vendor_f16(d.a, d.b, d.cref, M, N, K, o.layout, nullptr));
  LC_OKAY(lc_hgemm_f16(d.a, d.b, d.c, M, N, K, o.layout, o.variant, 2, o.stride, nullptr));
  HIP_OK(hipDeviceSynchronize());
  HIP_OK(hipMemcpy(hc.data(), d.c, hc.size() * 2, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(hr.data(), d.cref, hr.size() * 2, hipMemcpyDeviceToHost));
  float max_err = 0.f, worst_excess = -FLT_MAX;
  const float atol = 1e-3f + 2.5e-7f * (float)K + 2e-3f;   // tests/tol.py hgemm_atol + one fp16 ulp of the vendor's own rounding
  for (size_t i = 0; i < hc.size(); ++i) {
    const float r = h2f(hr[i]), c = h2f(hc[i]);
    const float err = fabsf(r - c);
    if (!(err == err)) { max_err = INFINITY; worst_excess = INFINITY; break; }
    max_err = fmaxf(max_err, err);
    worst_excess = fmaxf(worst_excess, err - (1e-2f * fabsf(r) + atol));
  }
  const bool ok = worst_excess <= 0.f;
  printf("M N K = %6d %6d %6d, Max Error = %f  (threshold |err| <= 1e-2*|ref| + %.4f: %s)\n", M, N, K, max_err, atol,
         ok ? "PASS" : "FAIL");
  return ok;
}

// perf_gemm: average seconds per launch (device events on the launch stream, inside lc_hgemm_time)
static double perf(const Opts& o, int M, int N, int K, DevBufs& d) {
  float ms = 0.f;
  LC_OKAY(lc_hgemm_time(d.a, d.b, d.c, M, N, K, o.layout, o.variant, 2, o.stride, o.warmup, o.inner, nullptr, &ms));
  return (double)ms * 1e-3;
}

int main(int argc, char** argv) {
  Opts o;
  for (int i = 1; i < argc; ++i) {
    auto is = [&](const char* s) { return strcmp(argv[i], s) == 0; };
    auto val = [&]() { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", argv[i]); exit(2); } return argv[++i]; };
    if (is("--layout")) o.layout = strcmp(val(), "tn") == 0 ? LC_LAYOUT_TN : LC_LAYOUT_NN;
    else if (is("--variant")) o.variant = atoi(val());
    else if (is("--max-n")) o.max_n = atoi(val());
    else if (is("--check-n")) o.check_n = atoi(val());
    else if (is("--outer")) o.outer = atoi(val());
    else if (is("--inner")) o.inner = atoi(val());
    else if (is("--warmup")) o.warmup = atoi(val());
    else if (is("--stride")) o.stride = atoi(val());
    else if (is("--mnk")) { o.M = atoi(val()); o.N = atoi(val()); o.K = atoi(val()); }
    else if (is("-h") || is("--help")) {
      printf("hgemm_bench [--layout nn|tn] [--variant N] [--max-n 16384] [--check-n 5] [--outer 10] [--inner 1] [--warmup 1] "
             "[--stride 2048] [--mnk M N K]\n");
      return 0;
    } else { fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
  }
  int cus = 0;
  LC_OKAY(lc_device_check(&cus));
  int diag = 0;
  printf("ALGO = MI355X MFMA HGEMM %s, variant %d (0 = auto), BLOCK SWIZZLE=%d, %d CUs, library: %s\n",
         o.layout == LC_LAYOUT_TN ? "TN" : "NN", o.variant, o.stride, cus, lc_build_info(&diag));
  if (diag) { fprintf(stderr, "refusing a LC_DIAG=1 library\n"); return 5; }

  std::vector<int> sizes;
  if (o.M > 0) sizes.push_back(-1);
  else for (int n = 256; n <= o.max_n; n += 256) sizes.push_back(n);   // hgemm_mma_stage.cu:1975-1979

  bool all_ok = true;
  for (int j = 0; j < o.check_n && j < (int)sizes.size(); ++j) {         // hgemm_mma_stage.cu:2003-2011
    const int M = o.M > 0 ? o.M : sizes[j], N = o.M > 0 ? o.N : sizes[j], K = o.M > 0 ? o.K : sizes[j];
    all_ok &= error_check(o, M, N, K);
  }
  (void)lc_vendor_destroy();

  for (int n : sizes) {                                                  // hgemm_mma_stage.cu:2013-2034
    const int M = n < 0 ? o.M : n, N = n < 0 ? o.N : n, K = n < 0 ? o.K : n;
    std::vector<uint16_t> ha((size_t)M * K), hb((size_t)K * N);
    fill(ha, 1u + (unsigned)M);
    fill(hb, 7u + (unsigned)N);
    DevBufs d(ha.size(), hb.size(), (size_t)M * N);
    HIP_OK(hipMemcpy(d.a, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d.b, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
    double max_sec = 0.0, min_sec = DBL_MAX, total = 0.0;
    for (int k = 0; k < o.outer; ++k) {
      const double s = perf(o, M, N, K, d);
      max_sec = fmax(max_sec, s);
      min_sec = fmin(min_sec, s);
      total += s;
    }
    const double avg = total / o.outer;
    printf("M N K = %6d %6d %6d, Time = %12.8lf %12.8lf %12.8lf s, AVG Performance = %10.4lf Tflops\n", M, N, K, min_sec,
           avg, max_sec, (double)M * N * K * 2 * 1e-12 / avg);
    fflush(stdout);
  }
  return all_ok ? 0 : 1;
}
