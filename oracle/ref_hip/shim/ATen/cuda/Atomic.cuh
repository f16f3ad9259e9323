// TEST INFRASTRUCTURE ONLY (oracle/_ref build): gpuAtomicAdd under its CUDA header name.
#pragma once
#include <ATen/hip/Atomic.cuh>
