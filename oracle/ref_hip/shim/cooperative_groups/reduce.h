// TEST INFRASTRUCTURE ONLY (oracle/_ref build): cg::reduce / cg::plus / cg::greater, which HIP's cooperative groups
// (ROCm 7.2) do not provide.  A tile-wide all-reduce as an xor butterfly over the tile (offsets size/2 … 1), the
// scheme of the CUDA header for static tiles; every lane returns the result.  cg::greater returns the larger value.
#pragma once
#include <hip/hip_cooperative_groups.h>
namespace cooperative_groups {
    template <typename T>
    struct plus {
        __device__ T operator()(T a, T b) const { return a + b; }
    };
    template <typename T>
    struct greater {
        __device__ T operator()(T a, T b) const { return a > b ? a : b; }
    };
    template <typename T>
    struct less {
        __device__ T operator()(T a, T b) const { return a < b ? a : b; }
    };
    template <unsigned int SIZE, class ParentT, typename T, class Op>
    __device__ inline T reduce(const thread_block_tile<SIZE, ParentT>& g, T val, Op op) {
#pragma unroll
        for (unsigned int offset = SIZE / 2; offset > 0; offset /= 2)
            val = op(val, g.shfl_xor(val, offset));
        return val;
    }
}  // namespace cooperative_groups
