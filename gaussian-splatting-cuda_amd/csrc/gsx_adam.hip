// gsx_adam.hip — fused Adam step.  Replaces fast_gs::optimizer::adam_step_wrapper
// (reference: fastgs/optimizer/include/adam_kernels.cuh:13-38, host logic src/training/optimizers/fused_adam.cpp:20-96).
// Pure streaming: 28 B per element (param r/w, exp_avg r/w, exp_avg_sq r/w, grad r) — HBM-bound.
// `rows x cols` with leading dimensions lets one launch update a strided view (the sh0 / shN column blocks of the
// single [N,K,3] SH tensor keep their own learning rate and state, like the reference's two parameter groups).
#include <algorithm>

#include "gsx_device.hpp"
#include "../../include/gsx.h"

namespace gsx {

void set_error(const char* msg);
int check_launch(const char* what);

__global__ __launch_bounds__(256) void adam_step_kernel(uint64_t rows, uint32_t cols, uint64_t ld_param, uint64_t ld_grad,
                                                        float* __restrict__ param, float* __restrict__ exp_avg,
                                                        float* __restrict__ exp_avg_sq, const float* __restrict__ grad, float lr,
                                                        float beta1, float beta2, float eps, float bc1_rcp, float bc2_sqrt_rcp) {
    const uint64_t total = rows * cols;
    const float step_size = lr * bc1_rcp;
    for (uint64_t idx = (uint64_t)blockIdx.x * 256u + threadIdx.x; idx < total; idx += (uint64_t)gridDim.x * 256u) {
        const uint64_t r = idx / cols, c = idx - r * cols;
        const float g = grad[r * ld_grad + c];
        const float m1 = beta1 * exp_avg[idx] + (1.0f - beta1) * g;
        const float m2 = beta2 * exp_avg_sq[idx] + (1.0f - beta2) * g * g;
        const float denom = sqrtf(m2) * bc2_sqrt_rcp + eps;
        param[r * ld_param + c] -= step_size * m1 / denom;
        exp_avg[idx] = m1;
        exp_avg_sq[idx] = m2;
    }
}

// Streaming accesses: the optimizer touches every byte once per step — nontemporal loads / stores keep the 1.6 GB of a 1 M-Gaussian step
// from displacing the lines the next render re-reads (measured: 59 floats x 1 M, six launches: 0.299 -> 0.276 ms = 6.0 TB/s).
// (nt_load4 / nt_store4: gsx_device.hpp)

// contiguous fast path: 16 B per lane per array
__global__ __launch_bounds__(256) void adam_step_vec4_kernel(uint64_t n4, float4* __restrict__ param, float4* __restrict__ exp_avg,
                                                             float4* __restrict__ exp_avg_sq, const float4* __restrict__ grad, float lr,
                                                             float beta1, float beta2, float eps, float bc1_rcp, float bc2_sqrt_rcp) {
    const float step_size = lr * bc1_rcp;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * 256u) {
        const float4 g = nt_load4(&grad[i]);
        float4 m = nt_load4(&exp_avg[i]), v = nt_load4(&exp_avg_sq[i]), p = nt_load4(&param[i]);
#define GSX_ADAM1(F)                                                        \
        m.F = beta1 * m.F + (1.0f - beta1) * g.F;                           \
        v.F = beta2 * v.F + (1.0f - beta2) * g.F * g.F;                     \
        p.F -= step_size * m.F / (sqrtf(v.F) * bc2_sqrt_rcp + eps);
        GSX_ADAM1(x) GSX_ADAM1(y) GSX_ADAM1(z) GSX_ADAM1(w)
#undef GSX_ADAM1
        nt_store4(p, &param[i]); nt_store4(m, &exp_avg[i]); nt_store4(v, &exp_avg_sq[i]);
    }
}

// Two column blocks of one dense [rows, cols] tensor (cols % 4 == 0), each with its own learning rate and an enable flag:
// the sh0 / shN parameter groups of the single [N,K,3] SH tensor in one fully vectorised launch.
__global__ __launch_bounds__(256) void adam_step_split_kernel(uint64_t n4, uint32_t cols, uint32_t split, float4* __restrict__ param,
                                                              float4* __restrict__ exp_avg, float4* __restrict__ exp_avg_sq,
                                                              const float4* __restrict__ grad, float step_a, float step_b, int do_a, int do_b,
                                                              float beta1, float beta2, float eps, float bc2_sqrt_rcp) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * 256u) {
        const uint32_t c = (uint32_t)((i * 4u) % cols);
        if (c + 3 < split ? !do_a : (c >= split ? !do_b : false)) continue;  // whole vector in a disabled block
        const float4 g = nt_load4(&grad[i]);
        float4 m = nt_load4(&exp_avg[i]), v = nt_load4(&exp_avg_sq[i]), p = nt_load4(&param[i]);
#define GSX_ADAM1(F, J)                                                                          \
        {                                                                                        \
            const bool a = c + J < split;                                                        \
            if (a ? do_a : do_b) {                                                               \
                m.F = beta1 * m.F + (1.0f - beta1) * g.F;                                        \
                v.F = beta2 * v.F + (1.0f - beta2) * g.F * g.F;                                  \
                p.F -= (a ? step_a : step_b) * m.F / (sqrtf(v.F) * bc2_sqrt_rcp + eps);          \
            }                                                                                    \
        }
        GSX_ADAM1(x, 0) GSX_ADAM1(y, 1) GSX_ADAM1(z, 2) GSX_ADAM1(w, 3)
#undef GSX_ADAM1
        nt_store4(p, &param[i]); nt_store4(m, &exp_avg[i]); nt_store4(v, &exp_avg_sq[i]);
    }
}

// Several dense tensors in ONE launch (the means / scaling / rotation / opacity groups of a training step: four launches of 3-16 MB
// each left the chip half idle between them).  Block b works on tensor t with blk_begin[t] <= b < blk_begin[t + 1].
struct AdamMulti {
    float* param[GSX_ADAM_MULTI_MAX]; float* exp_avg[GSX_ADAM_MULTI_MAX]; float* exp_avg_sq[GSX_ADAM_MULTI_MAX]; const float* grad[GSX_ADAM_MULTI_MAX];
    uint64_t n[GSX_ADAM_MULTI_MAX];
    float step[GSX_ADAM_MULTI_MAX], bc2[GSX_ADAM_MULTI_MAX];
    uint32_t blk_begin[GSX_ADAM_MULTI_MAX + 1];
    uint32_t count;
};

__global__ __launch_bounds__(256) void adam_step_multi_kernel(AdamMulti a, float beta1, float beta2, float eps) {
    uint32_t t = 0;
    while (t + 1 < a.count && blockIdx.x >= a.blk_begin[t + 1]) ++t;
    const uint32_t nb = a.blk_begin[t + 1] - a.blk_begin[t], b = blockIdx.x - a.blk_begin[t];
    float* __restrict__ P = a.param[t]; float* __restrict__ M = a.exp_avg[t]; float* __restrict__ V = a.exp_avg_sq[t];
    const float* __restrict__ G = a.grad[t];
    const float step_size = a.step[t], bc2 = a.bc2[t];
    const uint64_t n = a.n[t], n4 = n / 4;
    const bool vec = ((((uintptr_t)P | (uintptr_t)M | (uintptr_t)V | (uintptr_t)G) & 15u) == 0);
#define GSX_ADAM1(p, m, v, g)                                  \
    m = beta1 * m + (1.0f - beta1) * g;                        \
    v = beta2 * v + (1.0f - beta2) * g * g;                    \
    p -= step_size * m / (sqrtf(v) * bc2 + eps);
    if (vec) {
        for (uint64_t i = (uint64_t)b * 256u + threadIdx.x; i < n4; i += (uint64_t)nb * 256u) {
            const float4 g = nt_load4(reinterpret_cast<const float4*>(G) + i);
            float4 m = nt_load4(reinterpret_cast<float4*>(M) + i), v = nt_load4(reinterpret_cast<float4*>(V) + i), p = nt_load4(reinterpret_cast<float4*>(P) + i);
            GSX_ADAM1(p.x, m.x, v.x, g.x) GSX_ADAM1(p.y, m.y, v.y, g.y) GSX_ADAM1(p.z, m.z, v.z, g.z) GSX_ADAM1(p.w, m.w, v.w, g.w)
            nt_store4(p, reinterpret_cast<float4*>(P) + i); nt_store4(m, reinterpret_cast<float4*>(M) + i); nt_store4(v, reinterpret_cast<float4*>(V) + i);
        }
    }
    for (uint64_t i = (vec ? n4 * 4 : 0) + (uint64_t)b * 256u + threadIdx.x; i < n; i += (uint64_t)nb * 256u) {   // tail (or everything, unaligned)
        const float g = G[i];
        float m = M[i], v = V[i], p = P[i];
        GSX_ADAM1(p, m, v, g)
        P[i] = p; M[i] = m; V[i] = v;
    }
#undef GSX_ADAM1
}

}  // namespace gsx

using namespace gsx;

extern "C" int gsx_adam_step_multi(uint32_t count, float* const* param, float* const* exp_avg, float* const* exp_avg_sq,
                                   const float* const* grad, const uint64_t* n, const float* lr, const float* bias_correction1_rcp,
                                   const float* bias_correction2_sqrt_rcp, float beta1, float beta2, float eps, void* stream) {
    if (count == 0) return GSX_OK;
    if (count > GSX_ADAM_MULTI_MAX || !param || !exp_avg || !exp_avg_sq || !grad || !n || !lr || !bias_correction1_rcp || !bias_correction2_sqrt_rcp) {
        set_error("adam_step_multi: null argument or more than GSX_ADAM_MULTI_MAX tensors");
        return GSX_ERR_INVALID_ARGUMENT;
    }
    AdamMulti a;
    a.count = 0;
    uint32_t blocks = 0;
    for (uint32_t t = 0; t < count; ++t) {
        if (n[t] == 0) continue;
        if (!param[t] || !exp_avg[t] || !exp_avg_sq[t] || !grad[t]) { set_error("adam_step_multi: null tensor"); return GSX_ERR_INVALID_ARGUMENT; }
        const uint32_t k = a.count++;
        a.param[k] = param[t]; a.exp_avg[k] = exp_avg[t]; a.exp_avg_sq[k] = exp_avg_sq[t]; a.grad[k] = grad[t];
        a.n[k] = n[t]; a.step[k] = lr[t] * bias_correction1_rcp[t]; a.bc2[k] = bias_correction2_sqrt_rcp[t];
        a.blk_begin[k] = blocks;
        blocks += (uint32_t)std::min<uint64_t>((n[t] / 4 + 255) / 256 + 1, 8192u);   // blocks in proportion to the tensor's size
    }
    if (a.count == 0) return GSX_OK;
    a.blk_begin[a.count] = blocks;
    hipLaunchKernelGGL(adam_step_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, beta1, beta2, eps);
    return check_launch("adam_step_multi");
}

extern "C" int gsx_adam_step_split(uint64_t rows, uint32_t cols, uint32_t split, float* param, float* exp_avg, float* exp_avg_sq,
                                   const float* grad, float lr_a, float lr_b, int step_a, int step_b, float beta1, float beta2, float eps,
                                   float bias_correction1_rcp, float bias_correction2_sqrt_rcp, void* stream) {
    if (rows == 0 || cols == 0 || (!step_a && !step_b)) return GSX_OK;
    if (!param || !exp_avg || !exp_avg_sq || !grad || split > cols) {
        set_error("adam_step_split: null pointer / split beyond cols");
        return GSX_ERR_INVALID_ARGUMENT;
    }
    if (cols % 4 != 0 || ((((uintptr_t)param | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq | (uintptr_t)grad) & 15u) != 0)) {
        set_error("adam_step_split: cols must be a multiple of 4 and the arrays 16-byte aligned");
        return GSX_ERR_UNSUPPORTED;
    }
    const uint64_t n4 = rows * cols / 4;
    // 16 Ki blocks measured best on MI355X (sweep 1 Ki .. 1 Mi: 5.35 / 5.85 / 5.39 TB/s at 1 Ki / 16 Ki / 1 Mi)
    const uint32_t grid = (uint32_t)std::min<uint64_t>((n4 + 255) / 256, 16384u);
    hipLaunchKernelGGL(adam_step_split_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, n4, cols, split, (float4*)param, (float4*)exp_avg,
                       (float4*)exp_avg_sq, (const float4*)grad, lr_a * bias_correction1_rcp, lr_b * bias_correction1_rcp, step_a, step_b, beta1,
                       beta2, eps, bias_correction2_sqrt_rcp);
    return check_launch("adam_step_split");
}

extern "C" int gsx_adam_step(uint64_t rows, uint32_t cols, uint64_t ld_param, uint64_t ld_grad, float* param, float* exp_avg,
                             float* exp_avg_sq, const float* grad, float lr, float beta1, float beta2, float eps,
                             float bias_correction1_rcp, float bias_correction2_sqrt_rcp, void* stream) {
    if (rows == 0 || cols == 0) return GSX_OK;
    if (!param || !exp_avg || !exp_avg_sq || !grad || ld_param < cols || ld_grad < cols) {
        set_error("adam_step: null pointer / leading dimension smaller than cols");
        return GSX_ERR_INVALID_ARGUMENT;
    }
    hipStream_t st = (hipStream_t)stream;
    const uint64_t total = rows * cols;
    const bool contiguous = (ld_param == cols && ld_grad == cols) || rows == 1;
    const bool aligned = ((((uintptr_t)param | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq | (uintptr_t)grad) & 15u) == 0) && (total % 4 == 0);
    if (contiguous && aligned) {
        const uint64_t n4 = total / 4;
        const uint32_t grid = (uint32_t)std::min<uint64_t>((n4 + 255) / 256, 16384u);
        hipLaunchKernelGGL(adam_step_vec4_kernel, dim3(grid), dim3(256), 0, st, n4, (float4*)param, (float4*)exp_avg, (float4*)exp_avg_sq,
                           (const float4*)grad, lr, beta1, beta2, eps, bias_correction1_rcp, bias_correction2_sqrt_rcp);
    } else {
        const uint32_t grid = (uint32_t)std::min<uint64_t>((total + 255) / 256, 256u * 32u);
        hipLaunchKernelGGL(adam_step_kernel, dim3(grid), dim3(256), 0, st, rows, cols, ld_param, ld_grad, param, exp_avg, exp_avg_sq, grad,
                           lr, beta1, beta2, eps, bias_correction1_rcp, bias_correction2_sqrt_rcp);
    }
    return check_launch("adam_step");
}
