// torch_shim.h — glue between the reference's `void f(torch::Tensor...)` entry-point convention and the
// C-ABI (include/lc_abi.h).  Mirrors the host behaviour of the reference wrappers:
//   dtype / shape checks and messages : kernels/hgemm/mma/basic/hgemm_mma_stage.cu:2048-2057,
//                                       kernels/flash-attn/utils/utils.h:137-147
//   "headdim not support!"            : kernels/flash-attn/mma/basic/flash_attn_mma_split_q.cu:793
// Differences, all supersets: launches go to PyTorch's CURRENT HIP stream OF THE TENSORS' DEVICE, with that device made
// current for the call (the reference uses the legacy default stream of whatever device is current), a non-GPU tensor is
// rejected loudly (the reference dereferences it), and so are a non-contiguous view (the reference compares sizes only and
// would read the base storage in the wrong order) and tensors spread over several devices.
#pragma once
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
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

#define LC_CHECK_CONTIGUOUS(T)                                                                                  \
  if (!(T).is_contiguous()) {                                                                                   \
    throw std::runtime_error("leetcuda_amd: tensor must be contiguous (a transposed / sliced view would be read " \
                             "in its base storage order; call .contiguous())");                                \
  }

// every tensor of one call on ONE device (round-4 verdict, structure #11): the launch goes to that device's current stream
#define LC_CHECK_SAME_DEVICE(T, REF)                                                                  \
  if ((T).get_device() != (REF).get_device()) {                                                       \
    throw std::runtime_error("leetcuda_amd: all tensors of a call must live on the same device");    \
  }

// Makes the tensors' device current for the duration of a call (kernel attributes, the CU count and the launch itself are
// per-device state inside the C-ABI library) and hands out THAT device's current stream.
struct LcDeviceScope {
  c10::hip::HIPGuardMasqueradingAsCUDA guard;
  int index;
  explicit LcDeviceScope(const torch::Tensor& t) : guard(t.device()), index(t.get_device()) {}
  void* stream() const {
    return static_cast<void*>(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(static_cast<c10::DeviceIndex>(index)).stream());
  }
};

inline void lc_throw_on_error(int status, const char* entry) {
  if (status == LC_OK) return;
  if (status == LC_ERR_HEADDIM) throw std::runtime_error("headdim not support!");
  if (status == LC_ERR_SHAPE) throw std::runtime_error("Tensor size mismatch!");
  throw std::runtime_error(std::string(entry) + ": " + lc_status_string(status));
}
