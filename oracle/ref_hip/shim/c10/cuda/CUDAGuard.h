// TEST INFRASTRUCTURE ONLY (oracle/_ref build): the CUDA spelling of torch's device guard on a ROCm wheel.
#pragma once
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <c10/cuda/CUDAStream.h>
namespace at::cuda {
    using OptionalCUDAGuard = c10::hip::OptionalHIPGuardMasqueradingAsCUDA;
}  // namespace at::cuda
