// TEST INFRASTRUCTURE ONLY (oracle/_ref build): the CUDA spelling of torch's caching allocator on a ROCm wheel.
#pragma once
#include <c10/hip/HIPCachingAllocator.h>
namespace c10::cuda::CUDACachingAllocator {
    inline c10::Allocator* get() { return c10::hip::HIPCachingAllocator::get(); }
}  // namespace c10::cuda::CUDACachingAllocator
