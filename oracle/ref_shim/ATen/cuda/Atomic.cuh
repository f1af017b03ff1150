// TEST INFRASTRUCTURE (oracle/_ref build only).  Not reference code.
// deform_conv_cuda_kernel.cu:76 includes <ATen/cuda/Atomic.cuh> for gpuAtomicAdd; PyTorch-ROCm ships the
// same header as <ATen/hip/Atomic.cuh>.
#pragma once
#include <ATen/hip/Atomic.cuh>
