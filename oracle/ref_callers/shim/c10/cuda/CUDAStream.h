// TEST INFRASTRUCTURE ONLY (oracle/build_ref_callers.sh): the CUDA spellings include/core/camera.hpp uses for its private copy stream, on a ROCm wheel.
#pragma once
#include <c10/hip/HIPStream.h>
namespace at::cuda {
    using CUDAStream = c10::hip::HIPStream;
    inline CUDAStream getCurrentCUDAStream() { return c10::hip::getCurrentHIPStream(); }
    inline CUDAStream getStreamFromPool(const bool isHighPriority = false, c10::DeviceIndex device = -1) { return c10::hip::getStreamFromPool(isHighPriority, device); }
}  // namespace at::cuda
