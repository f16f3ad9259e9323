// probe of butterfly_reduce16 (gsx_raster_fast.hip) against host sums
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
// reduce x[0..15] over the 64 lanes; on return lane 16*r+15 holds in z[j] the total of value 4*j + {0,2,1,3}[r].
// The swaps are issued through inline asm: with hipcc/ROCm 7.2 `r[0] + r[1]` on the result of
// __builtin_amdgcn_permlane{32,16}_swap compiles to `v_add v, vdst, vdst` (the second result is lost; see
// tools/butterfly_probe.hip).  `s_nop 1` = the two wait states a VALU write needs before v_permlane*_swap reads it.
__device__ __forceinline__ void butterfly_reduce16(float (&x)[16], float (&z)[4]) {
    asm volatile("s_nop 1\n\t"
                 "v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\t"
                 "v_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\t"
                 "v_permlane32_swap_b32 %8, %9\n\tv_permlane32_swap_b32 %10, %11\n\t"
                 "v_permlane32_swap_b32 %12, %13\n\tv_permlane32_swap_b32 %14, %15\n\t"
                 "s_nop 1"
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]),
                   "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]));
    float y[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) y[j] = x[2 * j] + x[2 * j + 1];
    asm volatile("s_nop 1\n\t"
                 "v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\t"
                 "v_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7\n\t"
                 "s_nop 1"
                 : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7]));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float v = y[2 * j] + y[2 * j + 1];
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));  // row_shr:1
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));  // row_shr:2
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));  // row_shr:4
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));  // row_shr:8
        z[j] = v;
    }
}

__global__ void k(const float* in, float* out) {
    const int lane = threadIdx.x;
    float x[16], z[4];
    for (int i = 0; i < 16; ++i) x[i] = in[i * 64 + lane];
    butterfly_reduce16(x, z);
    if ((lane & 15) == 15) {
        const int row = lane >> 4;
        const int k0 = (row == 1) ? 2 : (row == 2 ? 1 : row);
        for (int j = 0; j < 4; ++j) out[4 * j + k0] = z[j];
    }
}
int main() {
    float h[16 * 64], *d, *o, r[16];
    for (int i = 0; i < 16 * 64; ++i) h[i] = (float)(rand() % 1000);
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, 64);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
    hipMemcpy(r, o, 64, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 16; ++i) { float s = 0; for (int l = 0; l < 64; ++l) s += h[i * 64 + l]; if (s != r[i]) { ++bad; printf("value %d: got %.0f want %.0f\n", i, r[i], s); } }
    printf("butterfly_reduce16: %s\n", bad ? "MISMATCH" : "ok");
    return bad;
}
