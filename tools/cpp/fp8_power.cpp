// fp8_power.cpp — what do the two MX fp8 MFMA forms sustain on RANDOM e4m3 data at the board's power limit?
// (diagnosis tool, standalone: hipcc + HIP runtime only; the fp8 twin of mfma_power.cpp.)
//
// Round 3's verdict asked for the fp8 GEMM (config 5b) on `v_mfma_scale_f32_16x16x128_f8f6f4` instead of
// `v_mfma_scale_f32_32x32x64_f8f6f4`.  For fp16 the 16x16 form is 14 % cheaper per FLOP at the cap (half the accumulator registers
// moved per FLOP); for the MX forms the operand traffic per FLOP DOUBLES with 16x16x128 (A 8 + B 8 + C 4 registers
// per 65 536 FLOP against 8 + 8 + 16 per 131 072), so the answer is not obvious.  Every wave (one per SIMD, 256 CUs x 4) holds a 64 x 128
// fp32 accumulator tile in literal AGPRs and the A (64 x 128) / B (128 x 128) e4m3 fragments of one K = 128 slab in VGPRs and issues
// nothing but MFMAs, unit block scales:
//   mode 0: 32x32x64   2 x 4 blocks, 2 k-steps  (16 MFMAs per slab)
//   mode 1: 16x16x128  4 x 8 blocks, 1 k-step   (32 MFMAs per slab)
// Same FLOPs and operand bytes per slab; the TFLOP/s ratio on randn-quantised data is the energy-per-FLOP ratio at the cap,
// on zeros (--zero) the issue-rate ratio.   usage: fp8_power.bin [--zero] [--seconds S]
#include <hip/hip_runtime.h>

#include "../../leetcuda_amd/csrc/lc_common.h"   // LC_AGPR_ALL

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <type_traits>
#include <utility>
#include <vector>

typedef int i32x8 __attribute__((ext_vector_type(8)));

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)

template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (N > 0) {
    sfor<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}
template <int R>
__device__ __forceinline__ void acc_zero() { asm volatile("v_accvgpr_write_b32 a[%0], 0" ::"n"(R) : LC_AGPR_ALL); }
template <int R>
__device__ __forceinline__ void mx32(i32x8 a, i32x8 b, int sc) {
  asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 a[%3:%4], %0, %1, a[%3:%4], %2, %2 op_sel_hi:[0,0,0]" ::"v"(a), "v"(b), "v"(sc), "n"(R),
               "n"(R + 15)
               : LC_AGPR_ALL);
}
template <int R>
__device__ __forceinline__ void mx16(i32x8 a, i32x8 b, int sc) {
  asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 a[%3:%4], %0, %1, a[%3:%4], %2, %2 op_sel_hi:[0,0,0]" ::"v"(a), "v"(b), "v"(sc), "n"(R),
               "n"(R + 3)
               : LC_AGPR_ALL);
}
template <int R>
__device__ __forceinline__ float acc_read() {
  float x;
  asm volatile("s_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 %0, a[%1]" : "=v"(x) : "n"(R) : LC_AGPR_ALL);
  return x;
}

// src: [12 fragments][256 threads] i32x8: 0..3 = A, 4..11 = B
template <int MODE>
__global__ __launch_bounds__(256) void fp8_power_kernel(const i32x8* __restrict__ src, float* __restrict__ out, int iters) {
  const int tid = threadIdx.x;
  i32x8 fa[4], fb[8];
#pragma unroll
  for (int f = 0; f < 4; ++f) fa[f] = src[f * 256 + tid];
#pragma unroll
  for (int f = 0; f < 8; ++f) fb[f] = src[(4 + f) * 256 + tid];
  int sc = 0x7f7f7f7f;
  asm volatile("" : "+v"(sc));
  sfor<128>([&](auto r) { acc_zero<decltype(r)::value>(); });
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0) {
      // blocks (i, j), i = 0..1, j = 0..3 at a[16 (4 i + j)]; k-step ks uses A fragment 2 ks + i, B fragment 4 ks + j
      sfor<2>([&](auto kc) {
        sfor<8>([&](auto bc) {
          constexpr int ks = decltype(kc)::value, i = decltype(bc)::value >> 2, j = decltype(bc)::value & 3;
          mx32<16 * (4 * i + j)>(fa[2 * ks + i], fb[4 * ks + j], sc);
        });
      });
    } else {
      // blocks (i, j), i = 0..3, j = 0..7 at a[4 (8 i + j)]; A fragment i, B fragment j (K = 128 in one MFMA)
      sfor<32>([&](auto bc) {
        constexpr int i = decltype(bc)::value >> 3, j = decltype(bc)::value & 7;
        mx16<4 * (8 * i + j)>(fa[i], fb[j], sc);
      });
    }
  }
  float s = acc_read<0>() + acc_read<37>() + acc_read<127>();
  if (s == 123.456f) out[blockIdx.x * 256 + tid] = s;   // keep the accumulators alive
}

template <int MODE>
static double run(const i32x8* src, float* out, int grid, int iters, double seconds) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(fp8_power_kernel<MODE>, dim3(grid), dim3(256), 0, 0, src, out, iters);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(fp8_power_kernel<MODE>, dim3(grid), dim3(256), 0, 0, src, out, iters);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const int n = std::max(1, (int)(seconds * 1e3 / ms));
  CK(hipEventRecord(e0));
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(fp8_power_kernel<MODE>, dim3(grid), dim3(256), 0, 0, src, out, iters);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double flops = (double)n * grid * 4.0 * iters * (64.0 * 128.0 * 128.0 * 2.0);
  return flops / (ms * 1e-3) * 1e-12;
}

static uint8_t to_e4m3(float x) {   // round-to-nearest OCP e4m3fn of a float in (-448, 448)
  uint8_t sign = x < 0 ? 0x80 : 0;
  float a = fabsf(x);
  if (a < 1.0f / 1024) return sign;
  int e;
  float m = frexpf(a, &e);   // a = m 2^e, m in [0.5, 1)
  int be = e - 1 + 7;        // biased exponent of 1.xxx form
  if (be <= 0) {             // subnormal: units of 2^-9
    int q = (int)lrintf(a * 512.0f);
    return sign | (uint8_t)(q > 7 ? 8 : q);
  }
  int q = (int)lrintf((m * 2.0f - 1.0f) * 8.0f);
  if (q == 8) { q = 0; ++be; }
  if (be > 15 || (be == 15 && q == 7)) { be = 15; q = 6; }
  return sign | (uint8_t)(be << 3) | (uint8_t)q;
}

int main(int argc, char** argv) {
  bool zero = false;
  double seconds = 1.5;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--zero")) zero = true;
    else if (!strcmp(argv[i], "--seconds") && i + 1 < argc) seconds = atof(argv[++i]);
  }
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int grid = prop.multiProcessorCount, iters = 4096;
  std::vector<uint8_t> h(12 * 256 * 32);
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  for (auto& b : h) b = zero ? 0 : to_e4m3(nd(rng));
  i32x8* src;
  float* out;
  CK(hipMalloc(&src, h.size()));
  CK(hipMalloc(&out, (size_t)grid * 256 * 4));
  CK(hipMemcpy(src, h.data(), h.size(), hipMemcpyHostToDevice));
  printf("fp8_power: %s, %d CUs x 4 waves, %s e4m3 data, %.1f s per mode, unit block scales\n", prop.gcnArchName, grid,
         zero ? "zero" : "randn-quantised", seconds);
  for (int rep = 0; rep < 2; ++rep) {
    printf("  v_mfma_scale_f32_32x32x64_f8f6f4   %8.1f TFLOP/s\n", run<0>(src, out, grid, iters, seconds));
    printf("  v_mfma_scale_f32_16x16x128_f8f6f4  %8.1f TFLOP/s\n", run<1>(src, out, grid, iters, seconds));
  }
  return 0;
}
