// TEST INFRASTRUCTURE ONLY (oracle/_ref build): the CUDA spelling of torch's current-stream query on a ROCm wheel.
#pragma once
#include <c10/hip/HIPStream.h>
namespace at::cuda {
    using CUDAStream = c10::hip::HIPStream;  // converts implicitly to hipStream_t
    inline CUDAStream getCurrentCUDAStream() { return c10::hip::getCurrentHIPStream(); }
}  // namespace at::cuda
