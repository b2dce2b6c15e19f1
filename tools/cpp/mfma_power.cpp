// mfma_power.cpp — what does the matrix pipe of an MI355X sustain on RANDOM data at its power-limited clock, per MFMA
// shape?  (diagnosis tool, standalone: hipcc + HIP runtime only.)
//
// Every wave (one per SIMD, 256 CUs x 4) keeps a 64 x 128 fp32 accumulator tile and the A (64 x 64) / B (128 x 64)
// fp16 fragments of one K = 64 slab in registers and issues nothing but MFMAs for ~1 s of back-to-back launches:
//   mode 0: v_mfma_f32_32x32x16_f16   2 x 4 blocks, 4 k-steps  (32 MFMAs per slab)
//   mode 1: v_mfma_f32_16x16x32_f16   4 x 8 blocks, 2 k-steps  (64 MFMAs per slab)
//   mode 2 / 3: the same two with every k-step's fragments RE-READ from LDS (ds_read_b128, the traffic of a 128 x 128
//               wave tile scaled to this tile: 0.75 KiB per 32x32x16 MFMA) — the energy price of the operand reads.
//   + 4: LDS-DMA stream (8 x 1 KiB per slab and wave: the L2 -> LDS rate of a 256 x 256 x 64 GEMM tile);
//   + 8: the LDS reads ROTATE through the LDS area, so every k-step brings data the registers did not hold before.
// Same FLOPs and operand bytes per slab in all modes, so the TFLOP/s ratio is the energy-per-FLOP ratio of the
// instruction mix once the chip sits at its power limit (a kernel's throughput there is set by joules
// per FLOP, not by issue slots).  Data: N(0,1) fp16 ("randn") or zeros (--zero) — zero operands toggle nothing.
#include <hip/hip_runtime.h>

#include "../../leetcuda_amd/csrc/lc_common.h"   // LC_AGPR_ALL: the clobber list that makes hipcc ALLOCATE the AGPRs the asm uses

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <ctime>
#include <random>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)

template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}
// accumulators in LITERAL AGPRs (hipcc shuffles 128 live accumulator values through v_accvgpr copies otherwise)
template <int R>
__device__ __forceinline__ void acc_zero() { asm volatile("v_accvgpr_write_b32 a[%0], 0" ::"n"(R) : LC_AGPR_ALL); }
template <int R>
__device__ __forceinline__ void mfma32(half8 a, half8 b) {
  asm volatile("v_mfma_f32_32x32x16_f16 a[%2:%3], %0, %1, a[%2:%3]" ::"v"(a), "v"(b), "n"(R), "n"(R + 15) : LC_AGPR_ALL);
}
template <int R>
__device__ __forceinline__ void mfma16(half8 a, half8 b) {
  asm volatile("v_mfma_f32_16x16x32_f16 a[%2:%3], %0, %1, a[%2:%3]" ::"v"(a), "v"(b), "n"(R), "n"(R + 3) : LC_AGPR_ALL);
}
template <int R>
__device__ __forceinline__ float acc_read1() {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(x) : "n"(R) : LC_AGPR_ALL);
  return x;
}
template <int R>
__device__ __forceinline__ float acc_read() {
  float x;
  asm volatile("s_nop 15\n\ts_nop 15\n\tv_accvgpr_read_b32 %0, a[%1]" : "=v"(x) : "n"(R) : LC_AGPR_ALL);
  return x;
}

template <int OFF>
__device__ __forceinline__ half8 lds_read_asm(uint32_t addr) {   // asynchronous: retire with lds_wait6 / lds_wait12
  half8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ void lds_wait(half8& a, half8& b, half8& c, half8& d, half8& e, half8& f) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
}

// frag layout in `src`: [24 fragments][256 threads] half8; fragments 0..7 = A (k-step major), 8..23 = B.
// MODE bit 0: MFMA shape (0 = 32x32x16, 1 = 16x16x32); bit 1: every k-step's fragments re-read from LDS one k-step ahead
// (asm ds_read_b128 behind the MFMAs, like the GEMM); bit 2: 8 LDS-DMA pieces (global_load_lds dwordx4, 1 KiB each) per
// slab and wave from `gsrc` — the L2 -> LDS rate of a 256 x 256 x 64 tile; 8 workgroups share each source stream.
template <int MODE>
__global__ __launch_bounds__(256) void mfma_power_kernel(const half8* __restrict__ src, const char* __restrict__ gsrc,
                                                        size_t region_bytes, int nstreams, float* __restrict__ out,
                                                        unsigned long long* __restrict__ clk, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  half8* lds = reinterpret_cast<half8*>(smem_raw);
  char* ring = smem_raw + 32 * 256 * 16;   // 32 KiB DMA landing zone (never read); [24, 32) x 4 KiB: rotation room
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  constexpr bool SHAPE16 = (MODE & 1) != 0, RD = (MODE & 2) != 0, DMA = (MODE & 4) != 0, ROT = (MODE & 8) != 0,
                 SWAP = (MODE & 16) != 0, ALT = (MODE & 32) != 0, EPI = (MODE & 64) != 0;
  unsigned long long t0 = 0, r0 = 0;
  if (blockIdx.x == 0 && tid == 0) {
    t0 = __builtin_readcyclecounter();
    r0 = __builtin_amdgcn_s_memrealtime();
  }
  half8 fa[8], fb[16];
#pragma unroll
  for (int f = 0; f < 8; ++f) fa[f] = src[f * 256 + tid];
#pragma unroll
  for (int f = 0; f < 16; ++f) fb[f] = src[(8 + f) * 256 + tid];
#pragma unroll
  for (int f = 0; f < 8; ++f) lds[f * 256 + tid] = fa[f];
#pragma unroll
  for (int f = 0; f < 16; ++f) lds[(8 + f) * 256 + tid] = fb[f];
#pragma unroll
  for (int f = 0; f < 8; ++f) lds[(24 + f) * 256 + tid] = src[(24 + f) * 256 + tid];
  __syncthreads();
  const uint32_t la0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(smem_raw) + tid * 16;
  // DMA source: stream (blockIdx / 8 ... ) — the 8 workgroups with equal blockIdx % 32 sit on ONE XCD (blockIdx % 8) and
  // read the same bytes at about the same time (an A / B panel shared by 8 tiles of an XCD)
  const char* gp = gsrc + (size_t)(blockIdx.x % nstreams) * region_bytes + wave * 8192 + lane * 16;
  size_t goff = 0;
  static_for<256>([&](auto r) { acc_zero<decltype(r)::value>(); });
  for (int it = 0; it < iters; ++it) {
    const uint32_t la = la0 + (ROT ? (uint32_t)(it & 7) * 4096u : 0u), la2 = la + 65536;
    if constexpr (!SHAPE16) {
      static_for<4>([&](auto kc) {   // A fragments 2ks, 2ks+1; B fragments 4ks .. 4ks+3
        constexpr int ks = decltype(kc)::value, kn = (ks + 1) & 3;
        if constexpr (RD) {
          fa[2 * kn] = lds_read_asm<((2 * kn) & 15) * 4096>(((2 * kn) < 16) ? la : la2);
          fa[2 * kn + 1] = lds_read_asm<((2 * kn + 1) & 15) * 4096>(((2 * kn + 1) < 16) ? la : la2);
          static_for<4>([&](auto j) { fb[4 * kn + decltype(j)::value] = lds_read_asm<((8 + 4 * kn + decltype(j)::value) & 15) * 4096>(((8 + 4 * kn + decltype(j)::value) < 16) ? la : la2); });
        }
        static_for<8>([&](auto c) {
          constexpr int i = decltype(c)::value >> 2, j = decltype(c)::value & 3;
          if constexpr (SWAP) mfma32<16 * (4 * i + j)>(fb[4 * ks + j], fa[2 * ks + i]);   // the operand shared by 4 consecutive MFMAs is SrcB
          else mfma32<16 * (4 * i + j)>(fa[2 * ks + i], fb[4 * ks + j]);                  // ... is SrcA
          if constexpr (ALT) mfma32<128 + 16 * (4 * i + j)>(fb[4 * ks + (j ^ 1)], fa[2 * ks + i]);   // second accumulator set a[128:255], twice the MFMAs
          if constexpr (DMA && (decltype(c)::value & 3) == 3) {   // 2 pieces per k-step = 8 per slab
            constexpr int p = 2 * ks + (decltype(c)::value >> 2);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + goff + p * 1024),
                                             (__attribute__((address_space(3))) void*)(ring + wave * 8192 + p * 1024), 16, 0, 0);
          }
        });
        if constexpr (RD) lds_wait(fa[2 * kn], fa[2 * kn + 1], fb[4 * kn], fb[4 * kn + 1], fb[4 * kn + 2], fb[4 * kn + 3]);
      });
    } else {
      static_for<2>([&](auto kc) {   // A fragments 4ks .. 4ks+3; B fragments 8ks .. 8ks+7
        constexpr int ks = decltype(kc)::value, kn = ks ^ 1;
        if constexpr (RD) {
          static_for<4>([&](auto i) { fa[4 * kn + decltype(i)::value] = lds_read_asm<((4 * kn + decltype(i)::value) & 15) * 4096>(((4 * kn + decltype(i)::value) < 16) ? la : la2); });
          static_for<8>([&](auto j) { fb[8 * kn + decltype(j)::value] = lds_read_asm<((8 + 8 * kn + decltype(j)::value) & 15) * 4096>(((8 + 8 * kn + decltype(j)::value) < 16) ? la : la2); });
        }
        static_for<32>([&](auto c) {
          constexpr int i = decltype(c)::value >> 3, j = decltype(c)::value & 7;
          mfma16<4 * (8 * i + j)>(fa[4 * ks + i], fb[8 * ks + j]);
          if constexpr (DMA && (decltype(c)::value & 7) == 7) {   // 4 pieces per k-step = 8 per slab
            constexpr int p = 4 * ks + (decltype(c)::value >> 3);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + goff + p * 1024),
                                             (__attribute__((address_space(3))) void*)(ring + wave * 8192 + p * 1024), 16, 0, 0);
          }
        });
        if constexpr (RD) {
          lds_wait(fa[4 * kn], fa[4 * kn + 1], fa[4 * kn + 2], fa[4 * kn + 3], fb[8 * kn], fb[8 * kn + 1]);
          lds_wait(fb[8 * kn + 2], fb[8 * kn + 3], fb[8 * kn + 4], fb[8 * kn + 5], fb[8 * kn + 6], fb[8 * kn + 7]);
        }
      });
    }
    if constexpr (DMA) {
      goff += 32768;   // the workgroup's 4 waves x 8 KiB
      if (goff >= region_bytes) goff = 0;
    }
  }
  if constexpr (EPI) {   // a GEMM tile's epilogue: every accumulator register leaves for global memory (128 KiB per workgroup)
    float* o2 = out + (size_t)gridDim.x * 256 + (size_t)blockIdx.x * 128 * 256 + tid;
    static_for<128>([&](auto r) { o2[decltype(r)::value * 256] = acc_read1<decltype(r)::value>(); });
  }
  out[blockIdx.x * 256 + tid] = acc_read<0>() + acc_read<127>();
  if (blockIdx.x == 0 && tid == 0) {
    clk[0] = __builtin_readcyclecounter() - t0;
    clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}

struct Res {
  double tflops, ghz;
};
template <int MODE>
Res run(const half8* src, const char* gsrc, size_t region, int nstreams, float* out, unsigned long long* clk, int iters, int launches, int grid) {
  constexpr int LDS_BYTES = 32 * 256 * 16 + 32768;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_power_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < launches / 2; ++i)
    hipLaunchKernelGGL(mfma_power_kernel<MODE>, dim3(grid), dim3(256), LDS_BYTES, 0, src, gsrc, region, nstreams, out, clk, iters);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < launches; ++i)
    hipLaunchKernelGGL(mfma_power_kernel<MODE>, dim3(grid), dim3(256), LDS_BYTES, 0, src, gsrc, region, nstreams, out, clk, iters);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
  unsigned long long h[2];
  CK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
  const double flops = 2.0 * 64 * 128 * 64 * 4 * (double)grid * iters * launches * ((MODE & 32) ? 2 : 1);   // per wave: one slab per iteration
  return {flops / (ms * 1e-3) * 1e-12, (double)h[0] / (double)h[1] * 0.1};
}

static double now_s() {
  timespec ts;
  clock_gettime(CLOCK_REALTIME, &ts);
  return ts.tv_sec + ts.tv_nsec * 1e-9;
}

int main(int argc, char** argv) {
  bool zero = false, bits = false;
  int iters = 2000;
  double seconds = 1.5;
  const char* modes = "0,1,2,3,10,11,4,14,15";
  int grid_mult = 1;   // workgroups per CU per launch (a GEMM launches tiles / CUs of them, one after the other)
  int nstreams = 32, region_mib = 8, region_kib = 0;   // DMA source: 32 streams x 8 MiB = 256 MiB (the size of A + B at 8192^3), 8 workgroups per stream
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--zero")) zero = true;
    else if (!strcmp(argv[i], "--bits")) bits = true;
    else if (!strcmp(argv[i], "--iters") && i + 1 < argc) iters = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--seconds") && i + 1 < argc) seconds = atof(argv[++i]);
    else if (!strcmp(argv[i], "--modes") && i + 1 < argc) modes = argv[++i];
    else if (!strcmp(argv[i], "--grid-mult") && i + 1 < argc) grid_mult = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--streams") && i + 1 < argc) nstreams = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--region-mib") && i + 1 < argc) region_mib = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--region-kib") && i + 1 < argc) region_kib = atoi(argv[++i]);
  }
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int grid = prop.multiProcessorCount * grid_mult;
  const size_t period = (size_t)16 << 20;   // bytes of distinct random data; the DMA source tiles it
  std::vector<_Float16> h(period / 2);
  std::mt19937 rng(0);
  std::normal_distribution<float> nd(0.f, 1.f);
  for (auto& x : h) x = zero ? (_Float16)0.f : (_Float16)nd(rng);
  if (bits)   // uniformly random 16-bit patterns (what an uninitialised register might hold: huge exponents, NaN, inf)
    for (auto& x : h) {
      const uint16_t u = (uint16_t)rng();
      memcpy(&x, &u, 2);
    }
  half8* src;
  float* out;
  char* gsrc;
  unsigned long long* clk;
  const size_t region = region_kib ? (size_t)region_kib << 10 : (size_t)region_mib << 20;
  CK(hipMalloc(&src, 32 * 256 * 16));
  CK(hipMalloc(&out, (size_t)grid * 256 * 4 * 129));
  CK(hipMalloc(&clk, 16));
  CK(hipMalloc(&gsrc, region * nstreams));
  CK(hipMemcpy(src, h.data(), 32 * 256 * 16, hipMemcpyHostToDevice));
  for (size_t o = 0; o < region * nstreams; o += period)
    CK(hipMemcpy(gsrc + o, h.data(), std::min(period, region * nstreams - o), hipMemcpyHostToDevice));
  printf("mfma_power: %s, %d CUs x 4 waves, %d slabs per launch, %.1f s per mode, %s data\n", prop.gcnArchName, grid, iters, seconds,
         bits ? "random-bit" : zero ? "zero" : "randn");
  printf("DMA source: %d streams x %zu KiB; grid = %d workgroups\n", nstreams, region >> 10, grid);
  printf("mode bits: 1 = 16x16x32 (else 32x32x16), 2 = fragments re-read from LDS, 8 = ... from rotating LDS offsets, 4 = LDS-DMA stream\n");
  std::string ms(modes);
  size_t pos = 0;
  while (pos < ms.size()) {
    const size_t c = ms.find(',', pos);
    const int m = atoi(ms.substr(pos, c == std::string::npos ? std::string::npos : c - pos).c_str());
    pos = c == std::string::npos ? ms.size() : c + 1;
    Res r{0, 0};
    double t0 = 0, t1 = 0;
#define CASE(M)                                                                                        \
  case M: {                                                                                            \
    const Res q = run<M>(src, gsrc, region, nstreams, out, clk, iters, 40, grid);                                \
    const double per = 2.0 * 64 * 128 * 64 * 4 * (double)grid * iters * ((M & 32) ? 2 : 1) / (q.tflops * 1e12);             \
    const int launches = (int)(seconds / per / 1.5) + 1;                                               \
    t0 = now_s();                                                                                      \
    r = run<M>(src, gsrc, region, nstreams, out, clk, iters, launches, grid);                                    \
    t1 = now_s();                                                                                      \
  } break;
    switch (m) {
      CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(10) CASE(11) CASE(14) CASE(15) CASE(16) CASE(32) CASE(48) CASE(26) CASE(30) CASE(64) CASE(96)
      default: printf("mode %d: not instantiated\n", m); continue;
    }
    printf("MODE %2d [%s%s%s%s%s%s] %7.1f TFLOP/s @ %.2f GHz  t0=%.3f t1=%.3f\n", m, (m & 1) ? "16x16x32" : "32x32x16",
           (m & 8) ? " +LDS(rot)" : (m & 2) ? " +LDS" : "", (m & 4) ? " +DMA" : "", (m & 16) ? " SrcB-shared" : "", (m & 32) ? " 256acc" : "", (m & 64) ? " +epilogue" : "", r.tflops, r.ghz, t0, t1);
    fflush(stdout);
  }
  return 0;
}
