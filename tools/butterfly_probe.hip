// probe of butterfly_reduce16 (gsx_raster_fast.hip) against host sums
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
// (verbatim copy of the function in gsx_raster_fast.hip)
// reduce x[0..15] over the 64 lanes; every lane of quad q = (lane >> 2) & 3 of row r = lane >> 4 returns the total of value
// 4*q + {0,2,1,3}[r].
// The swaps are issued through inline asm: with hipcc/ROCm 7.2 `r[0] + r[1]` on the result of
// __builtin_amdgcn_permlane{32,16}_swap compiles to `v_add v, vdst, vdst` (the second result is lost; see
// tools/butterfly_probe.hip).  `s_nop 1` = the two wait states a VALU write needs before v_permlane*_swap reads it.
__device__ __forceinline__ float butterfly_reduce16(float (&x)[16]) {
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
    // Inside a row the halving continues (4 values x 16 lanes -> 1 value per lane): exchange distance 8 (row_ror:8) keeps the
    // pair selected by lane bit 3, distance 4 (ds_swizzle xor 4, LDS crossbar: no VALU slot) the value selected by lane bit 2,
    // then the quad is summed with two quad_perm adds.  12 VALU instead of 20 for the four 16-lane row sums.
    const uint32_t ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const bool hi8 = (ln & 8u) != 0u, hi4 = (ln & 4u) != 0u;
    const float v0 = y[0] + y[1], v1 = y[2] + y[3], v2 = y[4] + y[5], v3 = y[6] + y[7];
    float k0 = hi8 ? v2 : v0, k1 = hi8 ? v3 : v1;
    const float s0 = hi8 ? v0 : v2, s1 = hi8 ? v1 : v3;
    k0 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s0), 0x128, 0xf, 0xf, true));  // row_ror:8
    k1 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s1), 0x128, 0xf, 0xf, true));
    float k = hi4 ? k1 : k0;
    const float sd = hi4 ? k0 : k1;
    k += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, sd), 0x101F));                   // lane ^ 4
    k += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, k), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
    k += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, k), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
    return k;
}

__global__ void k(const float* in, float* out) {
    const int lane = threadIdx.x;
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = in[i * 64 + lane];
    const float total = butterfly_reduce16(x);
    if ((lane & 3) == 0) {
        const int row = lane >> 4;
        out[4 * ((lane >> 2) & 3) + ((row == 1) ? 2 : (row == 2 ? 1 : row))] = total;
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
