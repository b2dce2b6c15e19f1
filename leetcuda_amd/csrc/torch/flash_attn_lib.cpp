// flash_attn_lib.cpp — the `flash_attn_lib` PyTorch extension module (the name the reference bench loads,
// kernels/flash-attn/flash_attn_mma.py:222-228) exporting the entry points of
// kernels/flash-attn/pybind/flash_attn.cc:170-223 (26 + the 3 BUILD_FLASH_ATTN_MMA_OTHERS names) with
// identical names and signatures.  No device code here: every call forwards to the C-ABI.
#include "torch_shim.h"

namespace {

void attn_dispatch(const char* entry, torch::Tensor Q, torch::Tensor K, torch::Tensor V, torch::Tensor O,
                   int stages) {
  LC_CHECK_DTYPE(Q, torch::kHalf)  // Q [B,H,N,D]
  LC_CHECK_DTYPE(K, torch::kHalf)  // K [B,H,N,D]
  LC_CHECK_DTYPE(V, torch::kHalf)  // V [B,H,N,D] ([B,H,D,N] for the *_swizzle_qkv share/tiling_qk entries)
  LC_CHECK_DTYPE(O, torch::kHalf)  // O [B,H,N,D]
  LC_CHECK_CONTIGUOUS(Q)
  LC_CHECK_CONTIGUOUS(K)
  LC_CHECK_CONTIGUOUS(V)
  LC_CHECK_CONTIGUOUS(O)
  LC_CHECK_DEVICE(Q)
  LC_CHECK_DEVICE(K)
  LC_CHECK_DEVICE(V)
  LC_CHECK_DEVICE(O)
  LC_CHECK_SAME_DEVICE(K, Q)
  LC_CHECK_SAME_DEVICE(V, Q)
  LC_CHECK_SAME_DEVICE(O, Q)
  if (Q.dim() != 4 || K.dim() != 4 || V.dim() != 4 || O.dim() != 4)
    throw std::runtime_error("Tensor size mismatch!");
  const int B = Q.size(0), H = Q.size(1), N = Q.size(2), D = Q.size(3);
  if (K.numel() != Q.numel() || V.numel() != Q.numel() || O.numel() != Q.numel())
    throw std::runtime_error("Tensor size mismatch!");
  const LcDeviceScope dev(Q);
  const int rc = lc_attn_call(entry, Q.data_ptr(), K.data_ptr(), V.data_ptr(), O.data_ptr(), B, H, N, D,
                              stages, dev.stream());
  lc_throw_on_error(rc, entry);
}

}  // namespace

#define LC_ATTN5(func)                                                                               \
  void func(torch::Tensor Q, torch::Tensor K, torch::Tensor V, torch::Tensor O, int stages) {        \
    attn_dispatch(#func, Q, K, V, O, stages);                                                        \
  }

LC_ATTN5(flash_attn_mma_stages_split_kv)
LC_ATTN5(flash_attn_mma_stages_split_q)
LC_ATTN5(flash_attn_mma_stages_split_q_shared_kv)
LC_ATTN5(flash_attn_mma_stages_split_q_shared_qkv)
LC_ATTN5(flash_attn_mma_stages_split_q_tiling_qk)
LC_ATTN5(flash_attn_mma_stages_split_q_tiling_qkv)
LC_ATTN5(flash_attn_mma_stages_split_q_shared_kv_acc_f32)
LC_ATTN5(flash_attn_mma_stages_split_q_shared_qkv_acc_f32)
LC_ATTN5(flash_attn_mma_stages_split_q_tiling_qk_acc_f32)
LC_ATTN5(flash_attn_mma_stages_split_q_tiling_qkv_acc_f32)
LC_ATTN5(flash_attn_mma_stages_split_q_shared_kv_swizzle_q)
LC_ATTN5(flash_attn_mma_stages_split_q_shared_kv_swizzle_qk)
LC_ATTN5(flash_attn_mma_stages_split_q_shared_kv_swizzle_qkv)
LC_ATTN5(flash_attn_mma_stages_split_q_shared_qkv_swizzle_q)
LC_ATTN5(flash_attn_mma_stages_split_q_shared_qkv_swizzle_qk)
LC_ATTN5(flash_attn_mma_stages_split_q_shared_qkv_swizzle_qkv)
LC_ATTN5(flash_attn_mma_stages_split_q_tiling_qk_swizzle_q)
LC_ATTN5(flash_attn_mma_stages_split_q_tiling_qk_swizzle_qk)
LC_ATTN5(flash_attn_mma_stages_split_q_tiling_qk_swizzle_qkv)
LC_ATTN5(flash_attn_mma_stages_split_q_tiling_qkv_swizzle_q)
LC_ATTN5(flash_attn_mma_stages_split_q_tiling_qkv_swizzle_qk)
LC_ATTN5(flash_attn_mma_stages_split_q_tiling_qkv_swizzle_qkv)
LC_ATTN5(flash_attn_mma_stages_split_q_tiling_qkv_acc_f32_swizzle_q)
LC_ATTN5(flash_attn_mma_stages_split_q_tiling_qkv_acc_f32_swizzle_qk)
LC_ATTN5(flash_attn_mma_stages_split_q_tiling_qkv_acc_f32_swizzle_qkv)
void flash_attn_cute(torch::Tensor Q, torch::Tensor K, torch::Tensor V, torch::Tensor O) {
  attn_dispatch("flash_attn_cute", Q, K, V, O, 2);
}
LC_ATTN5(flash_attn_mma_stages_split_q_shared_qkv_Os2g)
LC_ATTN5(flash_attn_mma_stages_split_q_shared_kv_acc_f32_rr)
LC_ATTN5(flash_attn_mma_stages_split_q_shared_qkv_acc_f32_rr)

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  LC_TORCH_BINDING(flash_attn_mma_stages_split_kv)
  LC_TORCH_BINDING(flash_attn_mma_stages_split_q)
  LC_TORCH_BINDING(flash_attn_mma_stages_split_q_shared_kv)
  LC_TORCH_BINDING(flash_attn_mma_stages_split_q_shared_qkv)
  LC_TORCH_BINDING(flash_attn_mma_stages_split_q_tiling_qk)
  LC_TORCH_BINDING(flash_attn_mma_stages_split_q_tiling_qkv)
  LC_TORCH_BINDING(flash_attn_mma_stages_split_q_shared_kv_acc_f32)
  LC_TORCH_BINDING(flash_attn_mma_stages_split_q_shared_qkv_acc_f32)
  LC_TORCH_BINDING(flash_attn_mma_stages_split_q_tiling_qk_acc_f32)
  LC_TORCH_BINDING(flash_attn_mma_stages_split_q_tiling_qkv_acc_f32)
  LC_TORCH_BINDING(flash_attn_mma_stages_split_q_shared_kv_swizzle_q)
  LC_TORCH_BINDING(flash_attn_mma_stages_split_q_shared_kv_swizzle_qk)
  LC_TORCH_BINDING(flash_attn_mma_stages_split_q_shared_kv_swizzle_qkv)
  LC_TORCH_BINDING(flash_attn_mma_stages_split_q_shared_qkv_swizzle_q)
  LC_TORCH_BINDING(flash_attn_mma_stages_split_q_shared_qkv_swizzle_qk)
  LC_TORCH_BINDING(flash_attn_mma_stages_split_q_shared_qkv_swizzle_qkv)
  LC_TORCH_BINDING(flash_attn_mma_stages_split_q_tiling_qk_swizzle_q)
  LC_TORCH_BINDING(flash_attn_mma_stages_split_q_tiling_qk_swizzle_qk)
  LC_TORCH_BINDING(flash_attn_mma_stages_split_q_tiling_qk_swizzle_qkv)
  LC_TORCH_BINDING(flash_attn_mma_stages_split_q_tiling_qkv_swizzle_q)
  LC_TORCH_BINDING(flash_attn_mma_stages_split_q_tiling_qkv_swizzle_qk)
  LC_TORCH_BINDING(flash_attn_mma_stages_split_q_tiling_qkv_swizzle_qkv)
  LC_TORCH_BINDING(flash_attn_mma_stages_split_q_tiling_qkv_acc_f32_swizzle_q)
  LC_TORCH_BINDING(flash_attn_mma_stages_split_q_tiling_qkv_acc_f32_swizzle_qk)
  LC_TORCH_BINDING(flash_attn_mma_stages_split_q_tiling_qkv_acc_f32_swizzle_qkv)
  LC_TORCH_BINDING(flash_attn_cute)
  LC_TORCH_BINDING(flash_attn_mma_stages_split_q_shared_qkv_Os2g)
  LC_TORCH_BINDING(flash_attn_mma_stages_split_q_shared_kv_acc_f32_rr)
  LC_TORCH_BINDING(flash_attn_mma_stages_split_q_shared_qkv_acc_f32_rr)
}
