// probe.hip — tiny kernels that expose the hardware lane maps the production kernels rely on
// (MFMA operand / accumulator layouts and the ds_read_b64_tr_b16 transpose). Tests only.
#pragma once
#include "lc_common.h"

namespace lc {

// One wave. a: [16 rows][32 k], b: [16 cols][32 k] (both k-contiguous). d[row][col] by the documented map.
__global__ void probe_mfma16_kernel(const half_t* a, const half_t* b, float* d) {
  const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
  const half8_t af = *(const half8_t*)(a + i * 32 + g * 8);
  const half8_t bf = *(const half8_t*)(b + i * 32 + g * 8);
  f32x4_t c = {0.f, 0.f, 0.f, 0.f};
  c = mfma16(af, bf, c);
#pragma unroll
  for (int r = 0; r < 4; ++r) d[(g * 4 + r) * 16 + i] = c[r];  // row = 4*(lane>>4)+r, col = lane&15
}

// One wave. a: [32 rows][16 k], b: [32 cols][16 k]. d[row][col], row = (r&3)+8*(r>>2)+4*(lane>>5).
__global__ void probe_mfma32_kernel(const half_t* a, const half_t* b, float* d) {
  const int lane = threadIdx.x & 63, l32 = lane & 31, hi = lane >> 5;
  const half8_t af = *(const half8_t*)(a + l32 * 16 + hi * 8);
  const half8_t bf = *(const half8_t*)(b + l32 * 16 + hi * 8);
  f32x16_t c;
#pragma unroll
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = mfma32(af, bf, c);
#pragma unroll
  for (int r = 0; r < 16; ++r) d[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + l32] = c[r];
}

// One wave. src: 64 lanes x 4 u16 (lane-linear, 8 B per lane) copied to LDS verbatim; every lane then
// issues ds_read_b64_tr_b16 at its own 8-byte slot and dumps the 4 values it received.
__global__ void probe_tr16_kernel(const uint16_t* src, uint16_t* dst) {
  __shared__ __attribute__((aligned(16))) uint16_t buf[256];
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < 4; ++j) buf[lane * 4 + j] = src[lane * 4 + j];
  __syncthreads();
  const half4_t v = lds_tr16(&buf[lane * 4]);
  *(u32x2_t*)(dst + lane * 4) = __builtin_bit_cast(u32x2_t, v);  // raw 8 bytes, no per-element casts
}

}  // namespace lc
