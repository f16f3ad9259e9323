// TEST INFRASTRUCTURE ONLY (oracle/build_ref_callers.sh).  Python entry to the reference's OWN render call site — gs::training::rasterize
// (src/training/rasterization/rasterizer.cpp:46-437 with the autograd functions of rasterizer_autograd.cpp:12-391, both compiled unmodified) —
// so that tests/test_gpu_reference_callers.py can run it, forward and backward, over either backend it was linked against:
//   gsplat_ref_callers_gsx.so   compat/gsplat headers + libgsx_gsplat_backend.so   (the drop-in: what INTEGRATION.md tells a maintainer to do)
//   gsplat_ref_callers_ref.so   the reference's gsplat/ headers + its own kernels compiled for gfx950 (oracle/_ref/obj_gsplat_ref_hip)
// The tensors handed in are the raw parameters (leaves of the caller's autograd graph); the reference's SplatData getters activate them.
#include "rasterization/rasterizer.hpp"
#include <torch/extension.h>

static std::tuple<torch::Tensor, torch::Tensor, torch::Tensor> render(torch::Tensor means, torch::Tensor sh0, torch::Tensor shN, torch::Tensor scaling_raw,
                                                                      torch::Tensor rotation_raw, torch::Tensor opacity_raw, int sh_degree, torch::Tensor R,
                                                                      torch::Tensor T, float fx, float fy, float cx, float cy, int width, int height,
                                                                      torch::Tensor bg, torch::Tensor radial, torch::Tensor tangential, int camera_model) {
    gs::SplatData model(sh_degree, means, sh0, shN, scaling_raw, rotation_raw, opacity_raw, 1.0f);
    model.set_active_sh_degree(sh_degree);
    gs::Camera cam(R, T, fx, fy, cx, cy, radial, tangential, static_cast<gsplat::CameraModelType>(camera_model), "parity", "", width, height, 0);
    auto out = gs::training::rasterize(cam, model, bg);
    return {out.image, out.alpha, out.radii};
}

// the same call with `antialiased = true` (the reference's --antialiasing): the projection is asked for the compensation factors (rasterizer.cpp:181,241);
// on this path the glue never applies them
static std::tuple<torch::Tensor, torch::Tensor, torch::Tensor> render_antialiased(torch::Tensor means, torch::Tensor sh0, torch::Tensor shN, torch::Tensor scaling_raw,
                                                                                   torch::Tensor rotation_raw, torch::Tensor opacity_raw, int sh_degree, torch::Tensor R,
                                                                                   torch::Tensor T, float fx, float fy, float cx, float cy, int width, int height,
                                                                                   torch::Tensor bg) {
    gs::SplatData model(sh_degree, means, sh0, shN, scaling_raw, rotation_raw, opacity_raw, 1.0f);
    model.set_active_sh_degree(sh_degree);
    gs::Camera cam(R, T, fx, fy, cx, cy, torch::empty({0}), torch::empty({0}), gsplat::CameraModelType::PINHOLE, "parity", "", width, height, 0);
    auto out = gs::training::rasterize(cam, model, bg, 1.0f, false, true);
    return {out.image, out.alpha, out.radii};
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("render", &render, "gs::training::rasterize of the reference, RGB mode: (image [3,H,W], alpha [1,H,W], radii [N])");
    m.def("render_antialiased", &render_antialiased, "the same with antialiased = true");
}
