// TEST INFRASTRUCTURE (oracle/_ref build only).  Not reference code.
// The reference's deformable kernels (detectron2/layers/csrc/deformable/deform_conv_cuda_kernel.cu:72)
// include <c10/cuda/CUDAGuard.h> and use at::cuda::CUDAGuard / at::cuda::getCurrentCUDAStream().  This
// directory is put FIRST on the include path of oracle/build_ref.py:build_dcn(), so that the file compiles
// where it lies, as HIP, against PyTorch-ROCm's own guard / stream types -- no hipify pass, nothing written
// next to the reference sources.
#pragma once
#include <hip/hip_runtime.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

namespace at {
namespace cuda {
using CUDAGuard = at::hip::HIPGuardMasqueradingAsCUDA;
inline hipStream_t getCurrentCUDAStream() {
  return at::hip::getCurrentHIPStreamMasqueradingAsCUDA();
}
} // namespace cuda
} // namespace at
