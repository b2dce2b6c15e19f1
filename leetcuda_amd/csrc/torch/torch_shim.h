// torch_shim.h — glue between the reference's `void f(torch::Tensor...)` entry-point convention and the
// C-ABI (include/lc_abi.h).  Mirrors the host behaviour of the reference wrappers:
//   dtype / shape checks and messages : kernels/hgemm/mma/basic/hgemm_mma_stage.cu:2048-2057,
//                                       kernels/flash-attn/utils/utils.h:137-147
//   "headdim not support!"            : kernels/flash-attn/mma/basic/flash_attn_mma_split_q.cu:793
// Differences, all supersets: launches go to PyTorch's CURRENT HIP stream (the reference uses the
// legacy default stream), and a non-GPU tensor is rejected loudly (the reference dereferences it).
#pragma once
#include <c10/hip/HIPStream.h>
#include <torch/extension.h>
#include <torch/types.h>

#include <iostream>
#include <stdexcept>
#include <string>

#include "lc_abi.h"

#define LC_STRINGFY(str) #str
#define LC_TORCH_BINDING(func) m.def(LC_STRINGFY(func), &func, LC_STRINGFY(func));

#define LC_CHECK_DTYPE(T, th_type)                                   \
  if (((T).options().dtype() != (th_type))) {                        \
    std::cout << "Tensor Info:" << (T).options() << std::endl;       \
    throw std::runtime_error("values must be " #th_type);            \
  }

#define LC_CHECK_DEVICE(T)                                                                     \
  if (!(T).is_cuda()) {                                                                        \
    throw std::runtime_error("leetcuda_amd: tensor must live on the MI355X (no CPU path)");   \
  }

inline void* lc_current_stream() {
  return static_cast<void*>(c10::hip::getCurrentHIPStream().stream());
}

inline void lc_throw_on_error(int status, const char* entry) {
  if (status == LC_OK) return;
  if (status == LC_ERR_HEADDIM) throw std::runtime_error("headdim not support!");
  if (status == LC_ERR_SHAPE) throw std::runtime_error("Tensor size mismatch!");
  throw std::runtime_error(std::string(entry) + ": " + lc_status_string(status));
}
