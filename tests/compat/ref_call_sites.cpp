// A translation unit shaped like the reference's callers of `namespace gsplat`: it includes the backend headers BY THE
// REFERENCE'S FILE NAMES ("Ops.h", "Projection.h", "Common.h", resolved through -I compat/gsplat), next to <torch/torch.h> as
// src/training/rasterization/*.cpp do, and calls every operator with the argument shapes of the reference call sites:
//   rasterizer.cpp:318 (intersect_tile with `{}` for the two optional id tensors), :327 (intersect_offset),
//   rasterizer_autograd.cpp:66 / :106 (SH fwd / bwd on flattened tensors), :217 (projection with std::nullopt and
//   std::optional<torch::Tensor> coefficient arguments), :290-292 / :364-367 (ut_params.to_tensor() into saved_data,
//   from_tensor() back, blend fwd / bwd with `std::optional(masks->contiguous())`-style optionals),
//   strategies/mcmc.cpp:153 (relocation), :360 (add_noise), strategies/default_strategy.cpp:96 (quats_to_rotmats).
// Linked against libgsx_gsplat_backend.so + libgsx.so only: no Python in the process.
//
//   ref_call_sites cpu   -> host-side checks (types, tensor round trip, CHECK_INPUT-style errors on CPU tensors)
//   ref_call_sites gpu   -> the whole chain on a small deterministic scene on cuda:0, prints checksums
#include <torch/torch.h>

#include "Common.h"
#include "Ops.h"
#include "Projection.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <optional>
#include <string>

namespace {

struct Settings {  // the fields of GUTRasterizationSettings that reach the operators (rasterizer_autograd.hpp:50-62)
    int width, height, tile_size;
    float eps2d = 0.3f, near_plane = 0.01f, far_plane = 10000.f, radius_clip = 0.f, scaling_modifier = 1.f;
    gsplat::CameraModelType camera_model = gsplat::CameraModelType::PINHOLE;
};

template <class F>
bool throws_c10_error(F&& f, const char* needle) {
    try {
        f();
    } catch (const c10::Error& e) {
        return std::strstr(e.what(), needle) != nullptr;
    }
    return false;
}

#define REQUIRE(cond)                                                         \
    do {                                                                      \
        if (!(cond)) {                                                        \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);     \
            return 1;                                                         \
        }                                                                     \
    } while (0)

struct Scene {
    torch::Tensor means, quats, scales, opacities, sh, viewmat, K;
};

// deterministic closed-form scene (the Python side of the test builds the same one)
Scene make_scene(int64_t N, int width, int height, torch::Device dev) {
    auto i = torch::arange(N, torch::kFloat64);
    auto x = torch::sin(i * 12.9898) * 0.9, y = torch::cos(i * 78.233) * 0.9, z = 2.0 + torch::frac(i * 0.61803398875) * 2.0;
    Scene s;
    s.means = torch::stack({x * z * 0.5, y * z * 0.5, z}, 1).to(torch::kFloat32);
    s.quats = torch::stack({torch::cos(i * 0.37) + 1.5, torch::sin(i * 1.1), torch::cos(i * 2.3), torch::sin(i * 0.7)}, 1).to(torch::kFloat32);
    s.scales = (0.01 + 0.04 * torch::stack({torch::frac(i * 0.137), torch::frac(i * 0.731), torch::frac(i * 0.377)}, 1)).to(torch::kFloat32);
    s.opacities = (0.3 + 0.6 * torch::frac(i * 0.2718)).to(torch::kFloat32);
    auto k = torch::arange(N * 16 * 3, torch::kFloat64).reshape({N, 16, 3});
    s.sh = (0.3 * (torch::frac(k * 0.0123457) - 0.5)).to(torch::kFloat32);
    s.viewmat = torch::eye(4, torch::kFloat32).unsqueeze(0);
    s.K = torch::tensor({{0.8f * width, 0.f, 0.5f * width}, {0.f, 0.8f * width, 0.5f * height}, {0.f, 0.f, 1.f}}).unsqueeze(0);
    for (auto* t : {&s.means, &s.quats, &s.scales, &s.opacities, &s.sh, &s.viewmat, &s.K}) *t = t->to(dev).contiguous();
    return s;
}

int run_cpu() {
    // types and constants the callers rely on
    static_assert(gsplat::PINHOLE == 0 && gsplat::ORTHO == 1 && gsplat::FISHEYE == 2, "CameraModelType values (Common.h:46-50)");
    static_assert(static_cast<int>(ShutterType::GLOBAL) == 4, "ShutterType order (Cameras.h:16-22)");
    static_assert(N_THREADS_PACKED == 256, "Common.h:52");
    REQUIRE(std::fabs(ALPHA_THRESHOLD - 1.f / 255.f) < 1e-12f);
    // saved_data round trip of the UT parameters (rasterizer_autograd.cpp:290, :364)
    UnscentedTransformParameters ut;
    ut.alpha = 0.25f; ut.kappa = 1.5f; ut.require_all_sigma_points_valid = false;
    c10::IValue saved = ut.to_tensor();
    const torch::Tensor t = saved.toTensor();
    REQUIRE(t.dim() == 1 && t.size(0) == 5 && t.scalar_type() == torch::kFloat32 && !t.is_cuda());
    const auto back = UnscentedTransformParameters::from_tensor(t);
    REQUIRE(back.alpha == 0.25f && back.beta == 2.f && back.kappa == 1.5f && back.in_image_margin_factor == 0.1f && !back.require_all_sigma_points_valid);
    REQUIRE(throws_c10_error([&] { UnscentedTransformParameters::from_tensor(torch::zeros({4})); }, "1D tensor of size 5"));
    c10::IValue cm = static_cast<int>(gsplat::CameraModelType::FISHEYE);  // ctx->saved_data["camera_model"] (:289, :362-363)
    REQUIRE(static_cast<gsplat::CameraModelType>(cm.toInt()) == gsplat::FISHEYE);
    // every operator rejects CPU tensors with the reference's CHECK_INPUT message (Common.h:12-17) as a c10::Error
    Settings st{64, 48, 16};
    const Scene s = make_scene(32, st.width, st.height, torch::kCPU);
    const auto none = std::optional<torch::Tensor>();
    REQUIRE(throws_c10_error([&] { gsplat::spherical_harmonics_fwd(3, s.means, s.sh, none); }, "must be a CUDA tensor"));
    REQUIRE(throws_c10_error([&] { gsplat::spherical_harmonics_bwd(16, 3, s.means, s.sh, none, s.means, true); }, "must be a CUDA tensor"));
    REQUIRE(throws_c10_error([&] { gsplat::intersect_tile(s.means.slice(1, 0, 2).unsqueeze(0).contiguous(), torch::zeros({1, 32, 2}, torch::kInt32),
                                                          s.opacities.unsqueeze(0), {}, {}, 1, 16, 4, 3, true); }, "must be a CUDA tensor"));
    REQUIRE(throws_c10_error([&] { gsplat::intersect_offset(torch::zeros({4}, torch::kInt64), 1, 4, 3); }, "must be a CUDA tensor"));
    REQUIRE(throws_c10_error([&] { gsplat::projection_ut_3dgs_fused(s.means, s.quats, s.scales, s.opacities, s.viewmat, std::nullopt, s.K, st.width, st.height,
                                                                    st.eps2d, st.near_plane, st.far_plane, st.radius_clip, false, st.camera_model, ut,
                                                                    ShutterType::GLOBAL, none, none, none); }, "must be a CUDA tensor"));
    REQUIRE(throws_c10_error([&] { gsplat::quats_to_rotmats(s.quats); }, "must be a CUDA tensor"));
    REQUIRE(throws_c10_error([&] { gsplat::relocation(s.opacities, s.scales, torch::ones({32}, torch::kInt32), torch::ones({51, 51}), 51); }, "must be a CUDA tensor"));
    REQUIRE(throws_c10_error([&] { gsplat::add_noise(s.opacities, s.scales, s.quats, s.means, s.means, 1e-3f); }, "must be a CUDA tensor"));
    std::printf("CPU OK\n");
    return 0;
}

int run_gpu() {
    REQUIRE(torch::cuda::is_available());
    const torch::Device dev(torch::kCUDA, 0);
    Settings st{160, 96, 16};
    const int64_t N = 3000;
    const Scene s = make_scene(N, st.width, st.height, dev);
    UnscentedTransformParameters ut_params;
    std::optional<torch::Tensor> radial_coeffs, tangential_coeffs, thin_prism_coeffs;  // undistorted pinhole: all empty, as rasterizer.cpp:183-195
    // projection (rasterizer_autograd.cpp:214-236)
    auto scaled_scales = s.scales * st.scaling_modifier;
    auto proj = gsplat::projection_ut_3dgs_fused(s.means, s.quats, scaled_scales, s.opacities, s.viewmat, std::nullopt, s.K, st.width, st.height, st.eps2d,
                                                 st.near_plane, st.far_plane, st.radius_clip, false, st.camera_model, ut_params, ShutterType::GLOBAL,
                                                 radial_coeffs, tangential_coeffs, thin_prism_coeffs);
    auto radii = std::get<0>(proj).contiguous();
    auto means2d = std::get<1>(proj).contiguous();
    auto depths = std::get<2>(proj).contiguous();
    REQUIRE(radii.sizes() == torch::IntArrayRef({1, N, 2}) && !std::get<4>(proj).defined());
    // SH colours (rasterizer.cpp:250-266, rasterizer_autograd.cpp:60-67)
    auto campos = torch::inverse(s.viewmat).index({torch::indexing::Slice(), torch::indexing::Slice(0, 3), 3});
    auto dirs = s.means.unsqueeze(0) - campos.unsqueeze(1);
    auto masks = (radii > 0).all(-1);
    auto coeffs = s.sh.unsqueeze(0).contiguous();
    auto dirs_flat = dirs.reshape({-1, 3});
    auto coeffs_flat = coeffs.reshape({-1, coeffs.size(-2), 3});
    auto masks_flat = masks.reshape({-1});
    auto colors = gsplat::spherical_harmonics_fwd(3, dirs_flat, coeffs_flat, masks_flat).reshape({1, N, 3}).contiguous();
    colors = torch::clamp_min(colors + 0.5f, 0.f) * masks.unsqueeze(-1);  // masked rows are unspecified upstream: zero them for the checksum
    // intersection (rasterizer.cpp:315-329)
    const int tile_width = (st.width + st.tile_size - 1) / st.tile_size, tile_height = (st.height + st.tile_size - 1) / st.tile_size;
    const auto isect_results = gsplat::intersect_tile(means2d, radii, depths, {}, {}, 1, st.tile_size, tile_width, tile_height, true);
    const auto tiles_per_gauss = std::get<0>(isect_results);
    const auto isect_ids = std::get<1>(isect_results);
    const auto flatten_ids = std::get<2>(isect_results);
    auto isect_offsets = gsplat::intersect_offset(isect_ids, 1, tile_width, tile_height);
    isect_offsets = isect_offsets.reshape({1, tile_height, tile_width});
    REQUIRE(tiles_per_gauss.sum().item<int64_t>() == flatten_ids.size(0) && flatten_ids.size(0) > 0);
    // blend forward (rasterizer_autograd.cpp:286-311)
    c10::IValue saved_ut = ut_params.to_tensor();
    auto bg_color = torch::full({1, 3}, 0.1f, s.means.options());
    std::optional<torch::Tensor> masks_opt;  // `masks.has_value() ? std::optional(masks->contiguous()) : std::nullopt`
    auto opac = s.opacities.unsqueeze(0).contiguous();
    auto results = gsplat::rasterize_to_pixels_from_world_3dgs_fwd(
        s.means.contiguous(), s.quats.contiguous(), scaled_scales.contiguous(), colors.contiguous(), opac.contiguous(), bg_color.contiguous(),
        masks_opt.has_value() ? std::optional(masks_opt->contiguous()) : std::nullopt, st.width, st.height, st.tile_size, s.viewmat, std::nullopt, s.K,
        st.camera_model, ut_params, ShutterType::GLOBAL, radial_coeffs, tangential_coeffs, thin_prism_coeffs, isect_offsets.contiguous(),
        flatten_ids.contiguous());
    auto rendered = std::get<0>(results), render_alpha = std::get<1>(results), last_ids = std::get<2>(results);
    REQUIRE(rendered.sizes() == torch::IntArrayRef({1, st.height, st.width, 3}) && last_ids.scalar_type() == torch::kInt32);
    // blend backward (rasterizer_autograd.cpp:362-371)
    auto ut_back = UnscentedTransformParameters::from_tensor(saved_ut.toTensor());
    auto v_render_colors = torch::ones_like(rendered) * 0.5f, v_render_alpha = torch::ones_like(render_alpha);
    auto raster_grads = gsplat::rasterize_to_pixels_from_world_3dgs_bwd(
        s.means, s.quats, scaled_scales, colors, opac, bg_color, masks_opt, st.width, st.height, st.tile_size, s.viewmat, std::nullopt, s.K,
        st.camera_model, ut_back, ShutterType::GLOBAL, radial_coeffs, tangential_coeffs, thin_prism_coeffs, isect_offsets, flatten_ids, render_alpha,
        last_ids, v_render_colors, v_render_alpha);
    auto v_colors = std::get<3>(raster_grads);
    // SH backward (rasterizer_autograd.cpp:104-110)
    auto sh_grads = gsplat::spherical_harmonics_bwd(16, 3, dirs_flat, coeffs_flat, masks_flat, v_colors.reshape({-1, 3}).contiguous(), true);
    // strategy operators (mcmc.cpp:153-158, :360-366, default_strategy.cpp:96)
    auto ratios = torch::full({N}, 2, torch::TensorOptions().dtype(torch::kInt32).device(dev));
    ratios = torch::clamp_max_(ratios, 51);
    auto binoms = torch::ones({51, 51}, s.means.options());
    auto relocation_result = gsplat::relocation(s.opacities, s.scales, ratios, binoms, 51);
    auto means_noised = s.means.clone();
    gsplat::add_noise(torch::logit(s.opacities).unsqueeze(-1).contiguous(), torch::log(s.scales), s.quats, torch::ones_like(s.means), means_noised, 1e-3f);
    const torch::Tensor rotmats = gsplat::quats_to_rotmats(s.quats);
    REQUIRE(rotmats.sizes() == torch::IntArrayRef({N, 3, 3}));
    torch::cuda::synchronize();
    auto sum = [](const torch::Tensor& t) { return t.to(torch::kFloat64).sum().item<double>(); };
    REQUIRE(std::isfinite(sum(rendered)) && std::isfinite(sum(std::get<0>(raster_grads))) && std::isfinite(sum(std::get<0>(sh_grads))));
    std::printf("GPU OK n_isects=%lld render=%.9g alpha=%.9g v_means_abs=%.9g v_scales_abs=%.9g v_coeffs_abs=%.9g new_opac=%.9g noised=%.9g rot=%.9g\n",
                (long long)flatten_ids.size(0), sum(rendered), sum(render_alpha), sum(std::get<0>(raster_grads).abs()), sum(std::get<2>(raster_grads).abs()),
                sum(std::get<0>(sh_grads).abs()), sum(std::get<0>(relocation_result)), sum(means_noised), sum(rotmats.abs()));
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    const std::string mode = argc > 1 ? argv[1] : "cpu";
    try {
        return mode == "gpu" ? run_gpu() : run_cpu();
    } catch (const std::exception& e) {
        std::printf("EXCEPTION: %s\n", e.what());
        return 2;
    }
}
