// lds_probe.hip — cycles per wave64 LDS instruction on gfx950, for the access shapes the blend kernels consider.
// One block; `waves` waves (1 = a single wave alone on the CU, 4 = one per SIMD, 8 = two per SIMD) hammer the LDS with the same
// instruction; reported: clock cycles per instruction per wave (s_memtime / 100 MHz-independent: uses wall_clock64 ticks -> cycles
// via the measured ratio of an s_sleep-free VALU loop is avoided: we report __builtin_readcyclecounter deltas).
//   hipcc --offload-arch=gfx950 -O3 tools/lds_probe.hip -o /tmp/lds_probe && /tmp/lds_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

constexpr int REP = 512;
typedef float vf2 __attribute__((ext_vector_type(2)));
typedef float vf4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void probe(unsigned long long* out, const int* perm) {
    __shared__ float s[16 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16 * 1024; i += blockDim.x) s[i] = 1.f;
    __syncthreads();
    const int slot = perm[threadIdx.x & 255];       // random slot in [0, 256)
    float acc = 0.f;
    float4 acc4 = make_float4(0, 0, 0, 0);
    volatile float* vs = s;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 8
    for (int r = 0; r < REP; ++r) {
        const int plane = (r & 15) * 256;
        if (MODE == 0) atomicAdd(&s[plane + (threadIdx.x & 255)], 1.f);            // ds_add_f32, one slot per lane (conflict free)
        if (MODE == 1) atomicAdd(&s[plane + slot], 1.f);                            // ds_add_f32, random slots
        if (MODE == 2) atomicAdd((unsigned*)&s[plane + (threadIdx.x & 255)], 1u);   // ds_add_u32
        if (MODE == 3) { vs[plane + slot] = vs[plane + slot] + 1.f; }               // plain read-modify-write, random slots
        if (MODE == 4) acc += vs[plane + wave * 16];                                // ds_read_b32, wave-uniform address
        if (MODE == 5) { const vf2 q = *(volatile vf2*)&s[plane + wave * 16]; acc += q.x + q.y; }   // ds_read_b64 uniform
        if (MODE == 6) { const vf4 q = *(volatile vf4*)&s[plane + wave * 16]; acc4.x += q.x; acc4.y += q.y; acc4.z += q.z; acc4.w += q.w; }  // b128 uniform
        if (MODE == 7) acc += vs[plane + slot];                                     // ds_read_b32 random slots
        if (MODE == 8) { vf4 w; w.x = acc4.x; w.y = acc4.y; w.z = acc4.z; w.w = acc4.w; *(volatile vf4*)&s[(plane + (threadIdx.x & 255) * 4) & 16383] = w; }       // ds_write_b128, one 16 B cell per lane
        if (MODE == 9) { const vf4 q = *(volatile vf4*)&s[(plane + (threadIdx.x & 255) * 4) & 16383]; acc4.x += q.x; acc4.w += q.w; }  // ds_read_b128 per-lane cells
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) out[wave] = t1 - t0;
    if (acc + acc4.x + acc4.y + acc4.z + acc4.w == 123456.f) out[63] = 1;
}

template <int MODE>
void run(const char* name, unsigned long long* d_out, const int* d_perm) {
    for (int waves : {1, 4, 8, 16}) {
        hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(64 * waves), 0, 0, d_out, d_perm);
        hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(64 * waves), 0, 0, d_out, d_perm);
        (void)hipDeviceSynchronize();
        unsigned long long h[64];
        (void)hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
        std::printf("%-44s waves=%2d  %7.1f cycles / instruction / wave   -> LDS pipe %6.1f cycles / instruction\n", name, waves, (double)h[0] / REP,
                    (double)h[0] / REP / waves);
    }
}

int main() {
    unsigned long long* d_out;
    int* d_perm;
    (void)hipMalloc(&d_out, 64 * 8);
    (void)hipMalloc(&d_perm, 256 * 4);
    std::vector<int> perm(256);
    unsigned x = 12345;
    for (int i = 0; i < 256; ++i) { x = x * 1664525u + 1013904223u; perm[i] = (x >> 8) & 255; }
    (void)hipMemcpy(d_perm, perm.data(), 256 * 4, hipMemcpyHostToDevice);
    run<0>("ds_add_f32 distinct slots", d_out, d_perm);
    run<1>("ds_add_f32 random slots", d_out, d_perm);
    run<2>("ds_add_u32 distinct slots", d_out, d_perm);
    run<3>("read + add + write (plain), random slots", d_out, d_perm);
    run<4>("ds_read_b32 wave-uniform address", d_out, d_perm);
    run<5>("ds_read_b64 wave-uniform address", d_out, d_perm);
    run<6>("ds_read_b128 wave-uniform address", d_out, d_perm);
    run<7>("ds_read_b32 random slots", d_out, d_perm);
    run<8>("ds_write_b128 per-lane cells", d_out, d_perm);
    run<9>("ds_read_b128 per-lane cells", d_out, d_perm);
    return 0;
}
