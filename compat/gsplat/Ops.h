// compat/gsplat/Ops.h — the operator surface of the reference's gsplat/Ops.h (lines 12-166: all ten free functions of
// `namespace gsplat`, same names, parameter order, types and return tuples), implemented by
// gaussian-splatting-cuda_amd/csrc/ops_shim.cpp on top of the C ABI of libgsx.so (include/gsx.h).
// A reference build puts `compat/gsplat` on its include path instead of `gsplat/` and links `libgsx_gsplat_backend.so`
// (+ libgsx.so) instead of its `gsplat_backend` static library: INTEGRATION.md.
#pragma once

#include <ATen/core/Tensor.h>
#include <c10/util/Optional.h>

#include <cstdint>
#include <tuple>

#include "Cameras.h"
#include "Common.h"

namespace gsplat {

// gsplat/Ops.h:12-25
at::Tensor spherical_harmonics_fwd(const uint32_t degrees_to_use, const at::Tensor dirs, const at::Tensor coeffs,
                                   const at::optional<at::Tensor> masks);
std::tuple<at::Tensor, at::Tensor> spherical_harmonics_bwd(const uint32_t K, const uint32_t degrees_to_use,
                                                           const at::Tensor dirs, const at::Tensor coeffs,
                                                           const at::optional<at::Tensor> masks,
                                                           const at::Tensor v_colors, bool compute_v_dirs);
// gsplat/Ops.h:28-43
std::tuple<at::Tensor, at::Tensor, at::Tensor> intersect_tile(const at::Tensor means2d, const at::Tensor radii,
                                                              const at::Tensor depths,
                                                              const at::optional<at::Tensor> camera_ids,
                                                              const at::optional<at::Tensor> gaussian_ids,
                                                              const uint32_t C, const uint32_t tile_size,
                                                              const uint32_t tile_width, const uint32_t tile_height,
                                                              const bool sort);
at::Tensor intersect_offset(const at::Tensor isect_ids, const uint32_t C, const uint32_t tile_width,
                            const uint32_t tile_height);
// gsplat/Ops.h:45-65
at::Tensor quats_to_rotmats(const at::Tensor quats);
std::tuple<at::Tensor, at::Tensor> relocation(at::Tensor opacities, at::Tensor scales, at::Tensor ratios, at::Tensor binoms,
                                              const int n_max);
void add_noise(at::Tensor raw_opacities, at::Tensor raw_scales, at::Tensor raw_quats, at::Tensor noise, at::Tensor means,
               const float current_lr);
// gsplat/Ops.h:69-98
std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor> projection_ut_3dgs_fused(
    const at::Tensor means, const at::Tensor quats, const at::Tensor scales, const at::optional<at::Tensor> opacities,
    const at::Tensor viewmats0, const at::optional<at::Tensor> viewmats1, const at::Tensor Ks,
    const uint32_t image_width, const uint32_t image_height, const float eps2d, const float near_plane,
    const float far_plane, const float radius_clip, const bool calc_compensations, const CameraModelType camera_model,
    const UnscentedTransformParameters ut_params, ShutterType rs_type, const at::optional<at::Tensor> radial_coeffs,
    const at::optional<at::Tensor> tangential_coeffs, const at::optional<at::Tensor> thin_prism_coeffs);
// gsplat/Ops.h:100-129
std::tuple<at::Tensor, at::Tensor, at::Tensor> rasterize_to_pixels_from_world_3dgs_fwd(
    const at::Tensor means, const at::Tensor quats, const at::Tensor scales, const at::Tensor colors,
    const at::Tensor opacities, const at::optional<at::Tensor> backgrounds, const at::optional<at::Tensor> masks,
    const uint32_t image_width, const uint32_t image_height, const uint32_t tile_size, const at::Tensor viewmats0,
    const at::optional<at::Tensor> viewmats1, const at::Tensor Ks, const CameraModelType camera_model,
    const UnscentedTransformParameters ut_params, ShutterType rs_type, const at::optional<at::Tensor> radial_coeffs,
    const at::optional<at::Tensor> tangential_coeffs, const at::optional<at::Tensor> thin_prism_coeffs,
    const at::Tensor tile_offsets, const at::Tensor flatten_ids);
// gsplat/Ops.h:131-166
std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor> rasterize_to_pixels_from_world_3dgs_bwd(
    const at::Tensor means, const at::Tensor quats, const at::Tensor scales, const at::Tensor colors,
    const at::Tensor opacities, const at::optional<at::Tensor> backgrounds, const at::optional<at::Tensor> masks,
    const uint32_t image_width, const uint32_t image_height, const uint32_t tile_size, const at::Tensor viewmats0,
    const at::optional<at::Tensor> viewmats1, const at::Tensor Ks, const CameraModelType camera_model,
    const UnscentedTransformParameters ut_params, ShutterType rs_type, const at::optional<at::Tensor> radial_coeffs,
    const at::optional<at::Tensor> tangential_coeffs, const at::optional<at::Tensor> thin_prism_coeffs,
    const at::Tensor tile_offsets, const at::Tensor flatten_ids, const at::Tensor render_alphas,
    const at::Tensor last_ids, const at::Tensor v_render_colors, const at::Tensor v_render_alphas);

}  // namespace gsplat
