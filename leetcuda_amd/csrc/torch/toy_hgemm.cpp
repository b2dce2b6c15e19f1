// toy_hgemm.cpp — the `toy_hgemm` PyTorch extension module (same name the reference bench imports first,
// kernels/hgemm/tools/utils.py:131) exporting the 38 entry points of kernels/hgemm/pybind/hgemm.cc:126-181
// with identical names and signatures.  No device code here: every call forwards to the C-ABI.
#include "torch_shim.h"

namespace {

void hgemm_dispatch(const char* entry, torch::Tensor a, torch::Tensor b, torch::Tensor c, int stages,
                    bool swizzle, int swizzle_stride) {
  LC_CHECK_DTYPE(a, torch::kHalf)
  LC_CHECK_DTYPE(b, torch::kHalf)
  LC_CHECK_DTYPE(c, torch::kHalf)
  LC_CHECK_CONTIGUOUS(a)
  LC_CHECK_CONTIGUOUS(b)   // (a TN operand is the CONTIGUOUS tensor utils.py:152-156 as_col_major returns, not a .t() view)
  LC_CHECK_CONTIGUOUS(c)
  LC_CHECK_DEVICE(a)
  LC_CHECK_DEVICE(b)
  LC_CHECK_DEVICE(c)
  LC_CHECK_SAME_DEVICE(b, a)
  LC_CHECK_SAME_DEVICE(c, a)
  if (a.dim() != 2 || b.dim() != 2 || c.dim() != 2) throw std::runtime_error("Tensor size mismatch!");
  const int M = a.size(0);
  const int K = a.size(1);
  const int N = b.size(1);  // TN entries still present B as a [K,N]-shaped tensor (utils.py:152-156)
  if (b.size(0) != K || c.size(0) != M || c.size(1) != N) throw std::runtime_error("Tensor size mismatch!");
  const LcDeviceScope dev(a);
  const int rc = lc_hgemm_call(entry, a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, stages,
                               swizzle ? 1 : 0, swizzle_stride, dev.stream());
  lc_throw_on_error(rc, entry);
}

}  // namespace

#define LC_HGEMM3(func) \
  void func(torch::Tensor a, torch::Tensor b, torch::Tensor c) { hgemm_dispatch(#func, a, b, c, 2, false, 1); }
#define LC_HGEMM6(func)                                                                          \
  void func(torch::Tensor a, torch::Tensor b, torch::Tensor c, int stages, bool swizzle,         \
            int swizzle_stride) {                                                                \
    hgemm_dispatch(#func, a, b, c, stages, swizzle, swizzle_stride);                             \
  }

// CUDA-core ladder names (kernels/hgemm/naive/hgemm.cu, hgemm_async.cu)
LC_HGEMM3(hgemm_naive_f16)
LC_HGEMM3(hgemm_sliced_k_f16)
LC_HGEMM3(hgemm_t_8x8_sliced_k_f16x4)
LC_HGEMM3(hgemm_t_8x8_sliced_k_f16x4_pack)
LC_HGEMM3(hgemm_t_8x8_sliced_k_f16x4_bcf)
LC_HGEMM3(hgemm_t_8x8_sliced_k_f16x4_pack_bcf)
LC_HGEMM3(hgemm_t_8x8_sliced_k_f16x8_pack_bcf)
LC_HGEMM3(hgemm_t_8x8_sliced_k_f16x8_pack_bcf_dbuf)
LC_HGEMM3(hgemm_t_8x8_sliced_k16_f16x8_pack_dbuf)
LC_HGEMM3(hgemm_t_8x8_sliced_k16_f16x8_pack_dbuf_async)
LC_HGEMM3(hgemm_t_8x8_sliced_k32_f16x8_pack_dbuf)
LC_HGEMM3(hgemm_t_8x8_sliced_k32_f16x8_pack_dbuf_async)
LC_HGEMM3(hgemm_t_16x8_sliced_k32_f16x8_pack_dbuf)
LC_HGEMM3(hgemm_t_16x8_sliced_k32_f16x8_pack_dbuf_async)
// vendor comparator (kernels/hgemm/cublas/hgemm_cublas.cu)
void init_cublas_handle() { lc_throw_on_error(lc_vendor_init(), "init_cublas_handle"); }
void destroy_cublas_handle() { lc_throw_on_error(lc_vendor_destroy(), "destroy_cublas_handle"); }
LC_HGEMM3(hgemm_cublas_tensor_op_nn)
LC_HGEMM3(hgemm_cublas_tensor_op_tn)
// WMMA names (kernels/hgemm/wmma)
LC_HGEMM3(hgemm_wmma_m16n16k16_naive)
LC_HGEMM3(hgemm_wmma_m16n16k16_mma4x2)
LC_HGEMM3(hgemm_wmma_m16n16k16_mma4x2_warp2x4)
LC_HGEMM3(hgemm_wmma_m16n16k16_mma4x2_warp2x4_dbuf_async)
LC_HGEMM3(hgemm_wmma_m32n8k16_mma2x4_warp2x4_dbuf_async)
LC_HGEMM6(hgemm_wmma_m16n16k16_mma4x2_warp2x4_stages)
LC_HGEMM6(hgemm_wmma_m16n16k16_mma4x2_warp2x4_stages_dsmem)
LC_HGEMM6(hgemm_wmma_m16n16k16_mma4x2_warp4x4_stages_dsmem)
LC_HGEMM6(hgemm_wmma_m16n16k16_mma4x4_warp4x4_stages_dsmem)
// MMA names (kernels/hgemm/mma)
LC_HGEMM3(hgemm_mma_m16n8k16_naive)
LC_HGEMM3(hgemm_mma_m16n8k16_mma2x4_warp4x4)
LC_HGEMM6(hgemm_mma_m16n8k16_mma2x4_warp4x4_stages)
LC_HGEMM6(hgemm_mma_m16n8k16_mma2x4_warp4x4_stages_dsmem)
LC_HGEMM6(hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem)
LC_HGEMM6(hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_x4)
LC_HGEMM6(hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_rr)
LC_HGEMM6(hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_swizzle)
LC_HGEMM6(hgemm_mma_m16n8k16_mma2x4_warp4x4_stages_dsmem_tn)
LC_HGEMM6(hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_tn_swizzle_x4)
LC_HGEMM6(hgemm_mma_stages_block_swizzle_tn_cute)

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  LC_TORCH_BINDING(hgemm_naive_f16)
  LC_TORCH_BINDING(hgemm_sliced_k_f16)
  LC_TORCH_BINDING(hgemm_t_8x8_sliced_k_f16x4)
  LC_TORCH_BINDING(hgemm_t_8x8_sliced_k_f16x4_pack)
  LC_TORCH_BINDING(hgemm_t_8x8_sliced_k_f16x4_bcf)
  LC_TORCH_BINDING(hgemm_t_8x8_sliced_k_f16x4_pack_bcf)
  LC_TORCH_BINDING(hgemm_t_8x8_sliced_k_f16x8_pack_bcf)
  LC_TORCH_BINDING(hgemm_t_8x8_sliced_k_f16x8_pack_bcf_dbuf)
  LC_TORCH_BINDING(hgemm_t_8x8_sliced_k16_f16x8_pack_dbuf)
  LC_TORCH_BINDING(hgemm_t_8x8_sliced_k16_f16x8_pack_dbuf_async)
  LC_TORCH_BINDING(hgemm_t_8x8_sliced_k32_f16x8_pack_dbuf)
  LC_TORCH_BINDING(hgemm_t_8x8_sliced_k32_f16x8_pack_dbuf_async)
  LC_TORCH_BINDING(hgemm_t_16x8_sliced_k32_f16x8_pack_dbuf)
  LC_TORCH_BINDING(hgemm_t_16x8_sliced_k32_f16x8_pack_dbuf_async)
  LC_TORCH_BINDING(init_cublas_handle)
  LC_TORCH_BINDING(destroy_cublas_handle)
  LC_TORCH_BINDING(hgemm_cublas_tensor_op_nn)
  LC_TORCH_BINDING(hgemm_cublas_tensor_op_tn)
  LC_TORCH_BINDING(hgemm_wmma_m16n16k16_naive)
  LC_TORCH_BINDING(hgemm_wmma_m16n16k16_mma4x2)
  LC_TORCH_BINDING(hgemm_wmma_m16n16k16_mma4x2_warp2x4)
  LC_TORCH_BINDING(hgemm_wmma_m16n16k16_mma4x2_warp2x4_dbuf_async)
  LC_TORCH_BINDING(hgemm_wmma_m32n8k16_mma2x4_warp2x4_dbuf_async)
  LC_TORCH_BINDING(hgemm_wmma_m16n16k16_mma4x2_warp2x4_stages)
  LC_TORCH_BINDING(hgemm_wmma_m16n16k16_mma4x2_warp2x4_stages_dsmem)
  LC_TORCH_BINDING(hgemm_wmma_m16n16k16_mma4x2_warp4x4_stages_dsmem)
  LC_TORCH_BINDING(hgemm_wmma_m16n16k16_mma4x4_warp4x4_stages_dsmem)
  LC_TORCH_BINDING(hgemm_mma_m16n8k16_naive)
  LC_TORCH_BINDING(hgemm_mma_m16n8k16_mma2x4_warp4x4)
  LC_TORCH_BINDING(hgemm_mma_m16n8k16_mma2x4_warp4x4_stages)
  LC_TORCH_BINDING(hgemm_mma_m16n8k16_mma2x4_warp4x4_stages_dsmem)
  LC_TORCH_BINDING(hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem)
  LC_TORCH_BINDING(hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_x4)
  LC_TORCH_BINDING(hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_rr)
  LC_TORCH_BINDING(hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_swizzle)
  LC_TORCH_BINDING(hgemm_mma_m16n8k16_mma2x4_warp4x4_stages_dsmem_tn)
  LC_TORCH_BINDING(hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_tn_swizzle_x4)
  LC_TORCH_BINDING(hgemm_mma_stages_block_swizzle_tn_cute)
}
