// TEST INFRASTRUCTURE ONLY (oracle/_ref build): force-included in front of the reference's .cu files so that the
// handful of CUDA runtime names they use resolve to their HIP equivalents.  The product never sees this file.
#pragma once
#include <hip/hip_runtime.h>
#define cudaSuccess hipSuccess
#define cudaFuncAttributeMaxDynamicSharedMemorySize hipFuncAttributeMaxDynamicSharedMemorySize
template <class F>
static inline hipError_t cudaFuncSetAttribute(F* fn, hipFuncAttribute attr, int value) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(fn), attr, value);
}
