// Does gfx950 skip the inactive 16-lane rows (or 32-lane halves) of a wave64 VALU instruction?  If it did, 4x4-block-granular exec masks
// would turn lane efficiency into time directly.  Runs the same unrolled instruction stream with different sets of active lanes.
// hipcc --offload-arch=gfx950 -O3 tools/exec_probe.hip -o gpurun_out/exec_probe && ./gpurun_out/exec_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP 256
template <int OP> __global__ void k(float* out, int iters, float seed, unsigned long long active) {
    float a0 = threadIdx.x * 1e-3f + seed, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float b = 1.0001f, c = 0.0001f;
    if ((active >> (threadIdx.x & 63)) & 1ull) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < REP / 8; ++r) {
                if (OP == 0) { asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)); }
                if (OP == 1) { asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)); }
                if (OP == 2) { asm volatile("v_min_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n v_min_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_min_f32 %6, %6, %8\n v_min_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)); }
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int OP> void run(const char* name, float* d, unsigned long long active) {
    const int iters = 2000, waves_per_simd = 4;
    const int blocks = 256 * waves_per_simd;
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.f, active);
    hipEventRecord(s);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.f, active);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    const double inst_per_simd = (double)iters * REP * waves_per_simd;
    printf("%-12s exec=%016llx  %.3f ms -> %.2f cycles per wave64 instruction per SIMD (at 2.4 GHz)\n", name, active, ms, ms * 1e-3 * 2.4e9 / inst_per_simd);
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    const unsigned long long masks[] = {~0ull, 0x00000000FFFFFFFFull, 0xFFFFFFFF00000000ull, 0x000000000000FFFFull, 0x0000FFFF0000FFFFull, 0x00000000FFFF0000ull,
                                        0x00FF00FF00FF00FFull, 0x1ull};
    for (unsigned long long m : masks) { run<0>("v_fma_f32", d, m); run<1>("v_exp_f32", d, m); run<2>("v_min_f32", d, m); }
    return 0;
}
