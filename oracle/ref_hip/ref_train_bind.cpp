// TEST INFRASTRUCTURE ONLY — never linked, imported or executed by the product path or a timed region.
//
// Python bindings of the REFERENCE's own training kernels, compiled for gfx950 where they lie by oracle/build_ref_train.sh into
// oracle/_ref/gsplat_ref_train.so: the fused SSIM kernels (src/training/kernels/ssim.cu) behind their header-only autograd wrapper
// (include/kernels/fused_ssim.cuh), and the fused Adam step (fastgs/optimizer/src/adam.cu + adam_api.cu, kernel in
// fastgs/optimizer/include/adam_kernels.cuh) that src/training/optimizers/fused_adam.cpp:20-96 calls per parameter group.
// Used by tests/ to pin gsx_ssim.hip / gsx_adam.hip against them on the same MI355X.
#include <torch/extension.h>

#include "adam_api.h"
#include "kernels/fused_ssim.cuh"

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "reference training kernels (fused SSIM, fused Adam) compiled for gfx950 (checker only)";
    m.def("fused_ssim", [](at::Tensor a, at::Tensor b, const std::string& padding, bool train) { return fused_ssim(a, b, padding, train); });
    m.def("fusedssim", [](double C1, double C2, at::Tensor a, at::Tensor b, bool train) { return fusedssim((float)C1, (float)C2, a, b, train); });
    m.def("fusedssim_backward", [](double C1, double C2, at::Tensor a, at::Tensor b, at::Tensor dL_dmap, at::Tensor dm1, at::Tensor ds1, at::Tensor ds12) {
        return fusedssim_backward((float)C1, (float)C2, a, b, dL_dmap, dm1, ds1, ds12);
    });
    m.def("adam_step", [](at::Tensor param, at::Tensor exp_avg, at::Tensor exp_avg_sq, at::Tensor grad, double lr, double beta1, double beta2, double eps,
                          double bias_correction1_rcp, double bias_correction2_sqrt_rcp) {
        fast_gs::optimizer::adam_step_wrapper(param, exp_avg, exp_avg_sq, grad, (float)lr, (float)beta1, (float)beta2, (float)eps, (float)bias_correction1_rcp,
                                              (float)bias_correction2_sqrt_rcp);
    });
}
