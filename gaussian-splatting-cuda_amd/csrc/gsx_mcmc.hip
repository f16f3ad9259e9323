// gsx_mcmc.hip — the three remaining functions of the reference's gsplat/Ops.h (Ops.h:45-65), needed so that
// libgsx can replace the whole `gsplat_backend` static library at link time (the MCMC / default densification
// strategies call them: src/training/strategies/mcmc.cpp, default_strategy.cpp:96).  SURVEY §8f rank 3.
//   quats_to_rotmats  gsplat/QuatToRotmatCUDA.cu:13-39   [N,4] wxyz -> [N,3,3] row-major
//   relocation        gsplat/RelocationCUDA.cu:11-43     Eq. (9) of "3D Gaussian Splatting as Markov Chain Monte Carlo"
//   add_noise         gsplat/RelocationCUDA.cu:88-141    means += lr * sigmoid(-100 (o - 0.005)) * Sigma * noise  (in place)
// All three are one-lane-per-Gaussian streaming kernels (HBM-bound, a few tens of bytes per Gaussian).
#include "gsx_device.hpp"

namespace gsx {

void set_error(const char* msg);
int check_launch(const char* what);

__global__ __launch_bounds__(256) void quats_to_rotmats_kernel(uint32_t N, const float* __restrict__ quats, float* __restrict__ rotmats) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= N) return;
    const float4 q = reinterpret_cast<const float4*>(quats)[i];
    const m33 R = quat_to_rotmat(q.x, q.y, q.z, q.w);
    float* o = rotmats + (size_t)i * 9;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) o[r * 3 + c] = R.a[r][c];
}

__global__ __launch_bounds__(256) void relocation_kernel(uint32_t N, const float* __restrict__ opacities, const float* __restrict__ scales,
                                                         const int32_t* __restrict__ ratios, const float* __restrict__ binoms, int n_max,
                                                         float* __restrict__ new_opacities, float* __restrict__ new_scales) {
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= N) return;
    const int n_idx = ratios[idx];
    const float op = opacities[idx];
    const float new_op = 1.0f - powf(1.0f - op, 1.0f / (float)n_idx);
    new_opacities[idx] = new_op;
    float denom_sum = 0.0f;
    for (int i = 1; i <= n_idx; ++i) {
        float sign = 1.f, p = new_op;  // (-1)^k and new_op^(k+1), advanced with k
        for (int k = 0; k <= i - 1; ++k) {
            const float term = (sign / sqrtf((float)(k + 1))) * p;
            denom_sum += binoms[(i - 1) * n_max + k] * term;
            sign = -sign;
            p *= new_op;
        }
    }
    const float coeff = op / denom_sum;
#pragma unroll
    for (int k = 0; k < 3; ++k) new_scales[(size_t)idx * 3 + k] = coeff * scales[(size_t)idx * 3 + k];
}

__global__ __launch_bounds__(256) void add_noise_kernel(uint32_t N, const float* __restrict__ raw_opacities, const float* __restrict__ raw_scales,
                                                        const float* __restrict__ raw_quats, const float* __restrict__ noise,
                                                        float* __restrict__ means, float current_lr) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= N) return;
    const float s2[3] = {__expf(2.f * raw_scales[(size_t)i * 3]), __expf(2.f * raw_scales[(size_t)i * 3 + 1]), __expf(2.f * raw_scales[(size_t)i * 3 + 2])};
    const float4 q = reinterpret_cast<const float4*>(raw_quats)[i];
    float w = q.x, x = q.y, y = q.z, z = q.w;
    const float inv = fminf(rsqrtf(x * x + y * y + z * z + w * w), 1e+12f);  // match torch normalize (RelocationCUDA.cu:91)
    const m33 R = quat_to_mat_raw(quat{w * inv, x * inv, y * inv, z * inv});
    const float nv[3] = {noise[(size_t)i * 3], noise[(size_t)i * 3 + 1], noise[(size_t)i * 3 + 2]};
    // covariance * noise = R S^2 R^T noise
    float t[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) t[k] = s2[k] * (R.a[0][k] * nv[0] + R.a[1][k] * nv[1] + R.a[2][k] * nv[2]);
    const float opacity = 1.f / (1.f + __expf(-raw_opacities[i]));
    const float op_sigmoid = 1.f / (1.f + __expf(100.f * opacity - 0.5f));
    const float f = current_lr * op_sigmoid;
#pragma unroll
    for (int r = 0; r < 3; ++r) means[(size_t)i * 3 + r] += f * (R.a[r][0] * t[0] + R.a[r][1] * t[1] + R.a[r][2] * t[2]);
}

}  // namespace gsx

using namespace gsx;

extern "C" int gsx_quats_to_rotmats(uint32_t N, const float* quats, float* rotmats, void* stream) {
    if (N == 0) return GSX_OK;
    if (!quats || !rotmats) { set_error("quats_to_rotmats: null pointer"); return GSX_ERR_INVALID_ARGUMENT; }
    hipLaunchKernelGGL(quats_to_rotmats_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, quats, rotmats);
    return check_launch("quats_to_rotmats");
}

extern "C" int gsx_relocation(uint32_t N, const float* opacities, const float* scales, const int32_t* ratios, const float* binoms,
                              int n_max, float* new_opacities, float* new_scales, void* stream) {
    if (N == 0) return GSX_OK;  // upstream skips the launch (RelocationCUDA.cu:63-66)
    if (!opacities || !scales || !ratios || !binoms || !new_opacities || !new_scales || n_max <= 0) {
        set_error("relocation: null pointer / n_max <= 0");
        return GSX_ERR_INVALID_ARGUMENT;
    }
    hipLaunchKernelGGL(relocation_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, opacities, scales, ratios, binoms,
                       n_max, new_opacities, new_scales);
    return check_launch("relocation");
}

extern "C" int gsx_add_noise(uint32_t N, const float* raw_opacities, const float* raw_scales, const float* raw_quats, const float* noise,
                             float* means, float current_lr, void* stream) {
    if (N == 0) return GSX_OK;
    if (!raw_opacities || !raw_scales || !raw_quats || !noise || !means) { set_error("add_noise: null pointer"); return GSX_ERR_INVALID_ARGUMENT; }
    hipLaunchKernelGGL(add_noise_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, raw_opacities, raw_scales, raw_quats,
                       noise, means, current_lr);
    return check_launch("add_noise");
}
