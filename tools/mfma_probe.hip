// v_mfma_f32_4x4x1_16B_f32 on gfx950: (1) which lane / register holds D[i][j] of the 16 4x4 outer products, (2) what an MFMA costs a
// VALU-bound wave: cycles per loop iteration for N_V independent v_fma + N_M MFMAs (4 waves per SIMD, every CU busy).
// hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o gpurun_out/mfma_probe && ./gpurun_out/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void layout(float* out) {
    const int lane = threadIdx.x;
    const float a = 1.f + lane, b = 100.f + lane;
    v4f c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];
}

template <int NV, int NM> __global__ void mix(float* out, int iters) {
    float x[16];
    for (int k = 0; k < 16; ++k) x[k] = threadIdx.x * 1e-3f + k;
    v4f acc[4];
    for (int k = 0; k < 4; ++k) acc[k] = v4f{0.f, 0.f, 0.f, 0.f};
    const float m = 1.0001f, d = 1e-4f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
            for (int k = 0; k < NV; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k % 16]) : "v"(m), "v"(d));
#pragma unroll
            for (int k = 0; k < NM; ++k) acc[k % 4] = __builtin_amdgcn_mfma_f32_4x4x1f32(x[k % 16], m, acc[k % 4], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int k = 0; k < 16; ++k) s += x[k];
    for (int k = 0; k < 4; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, int NM> void run(float* d) {
    const int iters = 2000, blocks = 256 * 4;
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(mix<NV, NM>), dim3(blocks), dim3(256), 0, 0, d, 10);
    hipEventRecord(s);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(mix<NV, NM>), dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    // per SIMD: 4 waves x iters x 4 reps of (NV + NM) instructions
    const double cyc = ms * 1e-3 * 2.4e9 / (4.0 * iters * 4);
    printf("NV=%2d NM=%2d  %.3f ms  -> %.1f cycles per (NV VALU + NM MFMA) group per wave-slot  (%.2f per instruction)\n", NV, NM, ms, cyc, cyc / (NV + NM));
}

int main() {
    float* d; hipMalloc(&d, 256 * 4 * 256 * 4);
    float h[256];
    hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int ok1 = 1, ok2 = 1;
    for (int lane = 0; lane < 64; ++lane)
        for (int r = 0; r < 4; ++r) {
            const int q = lane & ~3;
            const float h1 = (1.f + q + r) * (100.f + lane);   // D[r](lane) = A[quad lane r] * B[lane]
            const float h2 = (1.f + lane) * (100.f + q + r);   // D[r](lane) = A[lane] * B[quad lane r]
            ok1 &= h[lane * 4 + r] == h1; ok2 &= h[lane * 4 + r] == h2;
        }
    printf("layout: D[r](lane) = A[4*(lane/4)+r] * B[lane]: %s;  = A[lane] * B[4*(lane/4)+r]: %s   (lane 5: %g %g %g %g)\n", ok1 ? "YES" : "no", ok2 ? "YES" : "no",
           h[20], h[21], h[22], h[23]);
    run<16, 0>(d); run<16, 1>(d); run<16, 2>(d); run<16, 4>(d); run<16, 8>(d); run<8, 4>(d); run<0, 8>(d); run<32, 4>(d);
    return 0;
}
