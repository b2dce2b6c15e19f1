// lc_common.h — shared device/host helpers for the gfx950 kernels of libleetcuda_amd.so.
// gfx950 (MI355X, CDNA4) only: wave64, MFMA, LDS-DMA (global_load_lds_dwordx4), ds_read_b64_tr_b16.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lc {

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x4_raw_t __attribute__((__vector_size__(8)));  // operand type of the tr16 builtin

#define LC_LDS __attribute__((address_space(3)))
#define LC_GLOBAL __attribute__((address_space(1)))

#define LC_DEVINL __device__ __forceinline__

// Wave index as a provably wave-uniform SGPR value (threadIdx-derived values are divergent to hipcc).
LC_DEVINL int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// LDS-DMA: 64 lanes x 16 B = 1 KiB, written LANE-LINEARLY at `lds_wave_base + lane*16`.
// `lds_wave_base` must be wave-uniform; `gsrc` is per-lane (so swizzles go on the SOURCE address).
LC_DEVINL void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const LC_GLOBAL void*)gsrc, (LC_LDS void*)lds_wave_base, 16, 0, 0);
}

// 16-lane-group hardware transpose read: lane i of a group supplies the address of 4 consecutive
// halves = row (i>>2), cols 4*(i&3).. of a 4x16 block; lane i receives column i (4 rows).
LC_DEVINL half4_t lds_tr16(const void* lds_addr) {
  fp16x4_raw_t r = __builtin_amdgcn_ds_read_tr16_b64_v4f16((LC_LDS fp16x4_raw_t*)lds_addr);
  return __builtin_bit_cast(half4_t, r);
}

LC_DEVINL half8_t cat4(half4_t a, half4_t b) {
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

LC_DEVINL f32x4_t mfma16(half8_t a, half8_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
LC_DEVINL f32x16_t mfma32(half8_t a, half8_t b, f32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// 16-bit element flavour of a kernel: fragments travel as raw half8_t (16 bytes), only the MFMA opcode and
// the fp32 <-> 16-bit conversions differ between fp16 and bf16.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
template <bool BF16>
LC_DEVINL f32x16_t mfma32_16(half8_t a, half8_t b, f32x16_t c) {
  if constexpr (BF16)
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b),
                                                   c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
template <bool BF16>
LC_DEVINL half_t cvt16(float x) {   // round-to-nearest-even to fp16 or bf16, returned as raw 16 bits in a half_t
  if constexpr (BF16)
    return __builtin_bit_cast(half_t, (__bf16)x);
  else
    return (half_t)x;
}

// raw s_barrier the compiler may not move code across (no implied memory waits: pair it with explicit s_waitcnt)
LC_DEVINL void raw_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

// XCD-aware, bijective remap of the hardware block id: block b runs on XCD b%8 (observed, speed
// only); give every XCD a contiguous chunk of logical tile ids so neighbours share an L2.
LC_DEVINL int xcd_remap(int b, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = b & 7, idx = b >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

}  // namespace lc
