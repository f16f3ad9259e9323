// gsx_training_ops.h — link-level drop-ins for the two other CUDA operators the reference's `--gut` training step calls besides
// gsplat/Ops.h, with the reference's own names and signatures, implemented in
// gaussian-splatting-cuda_amd/csrc/ops_shim.cpp over the C ABI (include/gsx.h):
//   fast_gs::optimizer::adam_step_wrapper / adam_step   fastgs/optimizer/include/adam_api.h:11-21, adam.h:9-20
//   fusedssim / fusedssim_backward                       include/kernels/ssim.cuh:11-29
// (the header-only autograd wrapper include/kernels/fused_ssim.cuh of the reference works unchanged on top of these two).
#pragma once
#include <ATen/core/Tensor.h>

#include <tuple>

namespace fast_gs::optimizer {

void adam_step_wrapper(at::Tensor& param, at::Tensor& exp_avg, at::Tensor& exp_avg_sq, const at::Tensor& param_grad, const float lr,
                       const float beta1, const float beta2, const float eps, const float bias_correction1_rcp,
                       const float bias_correction2_sqrt_rcp);

// raw-pointer form (device pointers, current stream)
void adam_step(float* param, float* exp_avg, float* exp_avg_sq, const float* param_grad, const int n_elements, const float lr,
               const float beta1, const float beta2, const float eps, const float bias_correction1_rcp,
               const float bias_correction2_sqrt_rcp);

}  // namespace fast_gs::optimizer

std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor> fusedssim(float C1, float C2, at::Tensor& img1, at::Tensor& img2, bool train);

at::Tensor fusedssim_backward(float C1, float C2, at::Tensor& img1, at::Tensor& img2, at::Tensor& dL_dmap, at::Tensor& dm_dmu1,
                              at::Tensor& dm_dsigma1_sq, at::Tensor& dm_dsigma12);
