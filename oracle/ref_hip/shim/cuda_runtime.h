// TEST INFRASTRUCTURE ONLY (oracle/_ref builds): <cuda_runtime.h> as the reference's fastgs/utils/utils.h names it -> the HIP runtime.
#pragma once
#include <hip/hip_runtime.h>
#ifndef cudaSuccess
#define cudaSuccess hipSuccess
#endif
#define cudaDeviceSynchronize hipDeviceSynchronize
#define cudaGetErrorString hipGetErrorString
