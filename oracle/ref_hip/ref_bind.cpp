// TEST INFRASTRUCTURE ONLY — never linked, imported or executed by the product path or a timed region.
//
// Python bindings of the REFERENCE's own operators (namespace gsplat of /root/reference/gsplat/Ops.h), whose .cu/.cpp sources
// are compiled for gfx950 where they lie by oracle/build_ref_hip.sh into oracle/_ref/gsplat_ref_hip.so.  Used by tests/ to pin
// the CPU oracle and the HIP kernels against the reference's kernels executed on the same MI355X.
// Enums travel as ints: camera model 0 PINHOLE / 1 ORTHO / 2 FISHEYE (gsplat/Common.h), shutter 0..3 rolling, 4 GLOBAL
// (gsplat/Cameras.h); the UT parameters as the 5-float tensor of UnscentedTransformParameters::to_tensor().
#include <torch/extension.h>

#include "Ops.h"

namespace {

using T = at::Tensor;
using OT = at::optional<at::Tensor>;

UnscentedTransformParameters ut_of(const OT& t) {
    return t.has_value() ? UnscentedTransformParameters::from_tensor(t.value().cpu()) : UnscentedTransformParameters{};
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "reference gsplat operators compiled for gfx950 (checker only)";
    m.def("spherical_harmonics_fwd", &gsplat::spherical_harmonics_fwd);
    m.def("spherical_harmonics_bwd", &gsplat::spherical_harmonics_bwd);
    m.def("intersect_tile", &gsplat::intersect_tile);
    m.def("intersect_offset", &gsplat::intersect_offset);
    m.def("quats_to_rotmats", &gsplat::quats_to_rotmats);
    m.def("relocation", &gsplat::relocation);
    m.def("add_noise", &gsplat::add_noise);
    m.def("projection_ut_3dgs_fused",
          [](T means, T quats, T scales, OT opacities, T viewmats0, OT viewmats1, T Ks, uint32_t W, uint32_t H, float eps2d, float near_plane,
             float far_plane, float radius_clip, bool calc_compensations, int camera_model, OT ut, int rs_type, OT radial, OT tangential, OT thin_prism) {
              return gsplat::projection_ut_3dgs_fused(means, quats, scales, opacities, viewmats0, viewmats1, Ks, W, H, eps2d, near_plane, far_plane,
                                                      radius_clip, calc_compensations, static_cast<gsplat::CameraModelType>(camera_model), ut_of(ut),
                                                      static_cast<ShutterType>(rs_type), radial, tangential, thin_prism);
          });
    m.def("rasterize_to_pixels_from_world_3dgs_fwd",
          [](T means, T quats, T scales, T colors, T opacities, OT backgrounds, OT masks, uint32_t W, uint32_t H, uint32_t tile, T viewmats0,
             OT viewmats1, T Ks, int camera_model, OT ut, int rs_type, OT radial, OT tangential, OT thin_prism, T tile_offsets, T flatten_ids) {
              return gsplat::rasterize_to_pixels_from_world_3dgs_fwd(means, quats, scales, colors, opacities, backgrounds, masks, W, H, tile, viewmats0,
                                                                     viewmats1, Ks, static_cast<gsplat::CameraModelType>(camera_model), ut_of(ut),
                                                                     static_cast<ShutterType>(rs_type), radial, tangential, thin_prism, tile_offsets,
                                                                     flatten_ids);
          });
    m.def("rasterize_to_pixels_from_world_3dgs_bwd",
          [](T means, T quats, T scales, T colors, T opacities, OT backgrounds, OT masks, uint32_t W, uint32_t H, uint32_t tile, T viewmats0,
             OT viewmats1, T Ks, int camera_model, OT ut, int rs_type, OT radial, OT tangential, OT thin_prism, T tile_offsets, T flatten_ids,
             T render_alphas, T last_ids, T v_render_colors, T v_render_alphas) {
              return gsplat::rasterize_to_pixels_from_world_3dgs_bwd(means, quats, scales, colors, opacities, backgrounds, masks, W, H, tile, viewmats0,
                                                                     viewmats1, Ks, static_cast<gsplat::CameraModelType>(camera_model), ut_of(ut),
                                                                     static_cast<ShutterType>(rs_type), radial, tangential, thin_prism, tile_offsets,
                                                                     flatten_ids, render_alphas, last_ids, v_render_colors, v_render_alphas);
          });
}
