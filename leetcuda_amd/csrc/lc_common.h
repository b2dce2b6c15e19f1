// lc_common.h — shared device/host helpers for the gfx950 kernels of libleetcuda_amd.so.
// gfx950 (MI355X, CDNA4) only: wave64, MFMA, LDS-DMA (global_load_lds_dwordx4), ds_read_b64_tr_b16.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>

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

// LDS-DMA through a buffer descriptor: address = rsrc.base + soffset (SGPR, wave-uniform) + voffset (per lane, 32 bit) —
// no 64-bit per-lane address arithmetic per piece (buffer_load_dwordx4 ... offen lds).
typedef __amdgpu_buffer_rsrc_t buf_rsrc_t;
LC_DEVINL buf_rsrc_t make_rsrc(const void* base_wave_uniform) {
  // readfirstlane the pointer: a descriptor hipcc cannot PROVE wave-uniform gets a waterfall loop around every use
  const uint64_t p = (uint64_t)base_wave_uniform;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p), hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), (short)0, 0x7fffffff, 0x00020000);
}
LC_DEVINL void blds16(buf_rsrc_t rsrc, unsigned voffset, unsigned soffset, void* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LC_LDS void*)lds_wave_base, 16, (int)voffset, (int)soffset, 0, 0);
}

// 16-lane-group hardware transpose read: lane i of a group supplies the address of 4 consecutive
// halves = row (i>>2), cols 4*(i&3).. of a 4x16 block; lane i receives column i (4 rows).
LC_DEVINL half4_t lds_tr16(const void* lds_addr) {
  fp16x4_raw_t r = __builtin_amdgcn_ds_read_tr16_b64_v4f16((LC_LDS fp16x4_raw_t*)lds_addr);
  return __builtin_bit_cast(half4_t, r);
}

// The same transpose read as an asm statement.  Why: hipcc models the builtin as an LDS access that may alias an
// in-flight LDS-DMA (global_load_lds) and guards EVERY builtin transpose read issued after a DMA with
// s_waitcnt vmcnt(0) — draining the whole DMA prefetch at each K tile (plain ds_read_b128 loads are not guarded).
// An asm load is invisible to hipcc's s_waitcnt bookkeeping: the caller must execute lds_tr16_wait*() — which
// names the destinations — before the first use of the results (in-order LDS returns: lgkmcnt(0) covers them).
template <int OFF>
LC_DEVINL half4_t lds_tr16_asm(uint32_t lds_byte_addr) {
  half4_t r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(lds_byte_addr), "n"(OFF));
  return r;
}
LC_DEVINL uint32_t lds_addr32(const void* p) { return (uint32_t)(uintptr_t)(LC_LDS const char*)p; }
LC_DEVINL void lds_tr16_wait8(half4_t (&a)[8]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
}
LC_DEVINL void lds_tr16_wait16(half4_t (&a)[16]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                 "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]),
                 "+v"(a[15]));
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

// Overflow guards of the attention kernels ("is this row sum of exponentials still below the limit?") decided on the BIT
// PATTERN: every unit is compiled with -fno-honor-nans, under which the compiler may rewrite !(a < b) as a >= b or fold
// isfinite(), so a float compare is not a reliable route for NaN / inf into the slow path.  x is >= 0 or not a number:
// non-negative floats order like unsigned integers, and +inf, every NaN and anything with the sign bit set compare ABOVE
// every finite positive limit.
LC_DEVINL bool psum_below(float x, float limit) {
  return __builtin_bit_cast(uint32_t, x) < __builtin_bit_cast(uint32_t, limit);
}
LC_DEVINL bool finite_bits(float x) { return (__builtin_bit_cast(uint32_t, x) & 0x7f800000u) != 0x7f800000u; }

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

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N-1>{}) — indices usable as
// template arguments / "n" asm operands (literal AGPR numbers)
template <int... Is, typename F>
LC_DEVINL void static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
LC_DEVINL void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

}  // namespace lc

// clobber list naming every AGPR: statements that address accumulators literally (a[N:M]) carry it so that hipcc
// never parks a value of its own in the accumulator half of the register file
// (a translation unit whose kernels run TWO waves per SIMD defines a shorter list before including this header: a clobbered AGPR
// counts towards the kernel's register allocation; round 3's eight-wave D = 64 experiment did)
#ifndef LC_AGPR_ALL
#define LC_AGPR_ALL \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14",  \
  "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28",  \
  "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42",  \
  "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56",  \
  "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70",  \
  "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84",  \
  "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98",  \
  "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110",  \
  "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122",  \
  "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134",  \
  "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146",  \
  "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158",  \
  "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170",  \
  "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182",  \
  "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194",  \
  "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206",  \
  "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218",  \
  "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230",  \
  "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242",  \
  "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254",  \
  "a255"
#endif


// a[0:127] + a[192:255]: the attention kernel's O accumulators and Q fragments; a[128:191] stay with hipcc (it
// spills VGPRs there with v_accvgpr_write/read instead of scratch memory)
#define LC_AGPR_ATTN \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14",  \
  "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28",  \
  "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42",  \
  "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56",  \
  "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70",  \
  "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84",  \
  "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98",  \
  "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110",  \
  "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122",  \
  "a123", "a124", "a125", "a126", "a127", "a192", "a193", "a194", "a195", "a196", "a197", "a198",  \
  "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210",  \
  "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222",  \
  "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234",  \
  "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246",  \
  "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"

namespace lc {
// Sixteen accumulator registers -> sixteen VGPRs in ONE asm statement: four groups of four consecutive AGPRs (first registers B0 .. B3).
// Why one statement (round 5, tools/attn_w4u_stamps.py): an epilogue that reads accumulators one `asm volatile` at a time gets a pad behind
// every statement and — volatile statements keep their order — a serial read / scale / convert / pack chain on four or five registers: ~ 15
// dependent instructions per four values, 3600 – 4300 cycles for the 128 values of an attention block's wave, ~ 8000 for the 256 of a GEMM
// tile's.  Sixteen independent values per statement let hipcc interleave the conversions.  The caller has drained the MFMAs.
template <int B0, int B1, int B2, int B3>
LC_DEVINL void acc_read16(float (&x)[16]) {
  asm volatile("v_accvgpr_read_b32 %0, a[%16]\n\tv_accvgpr_read_b32 %1, a[%17]\n\tv_accvgpr_read_b32 %2, a[%18]\n\tv_accvgpr_read_b32 %3, a[%19]\n\t"
               "v_accvgpr_read_b32 %4, a[%20]\n\tv_accvgpr_read_b32 %5, a[%21]\n\tv_accvgpr_read_b32 %6, a[%22]\n\tv_accvgpr_read_b32 %7, a[%23]\n\t"
               "v_accvgpr_read_b32 %8, a[%24]\n\tv_accvgpr_read_b32 %9, a[%25]\n\tv_accvgpr_read_b32 %10, a[%26]\n\tv_accvgpr_read_b32 %11, a[%27]\n\t"
               "v_accvgpr_read_b32 %12, a[%28]\n\tv_accvgpr_read_b32 %13, a[%29]\n\tv_accvgpr_read_b32 %14, a[%30]\n\tv_accvgpr_read_b32 %15, a[%31]"
               : "=v"(x[0]), "=v"(x[1]), "=v"(x[2]), "=v"(x[3]), "=v"(x[4]), "=v"(x[5]), "=v"(x[6]), "=v"(x[7]), "=v"(x[8]), "=v"(x[9]), "=v"(x[10]),
                 "=v"(x[11]), "=v"(x[12]), "=v"(x[13]), "=v"(x[14]), "=v"(x[15])
               : "n"(B0), "n"(B0 + 1), "n"(B0 + 2), "n"(B0 + 3), "n"(B1), "n"(B1 + 1), "n"(B1 + 2), "n"(B1 + 3), "n"(B2), "n"(B2 + 1), "n"(B2 + 2),
                 "n"(B2 + 3), "n"(B3), "n"(B3 + 1), "n"(B3 + 2), "n"(B3 + 3)
               : LC_AGPR_ALL);
}
}  // namespace lc
