// gsx_ut_project.hpp — the per-Gaussian body of the 3DGUT unscented-transform projection (reference kernel
// gsplat/ProjectionUT3DGSFused.cu:74-203, sigma points / UT gsplat/Cameras.cuh:1028-1150), shared by projection_ut_kernel
// (gsx_projection.hip) and the fused front end (gsx_frontend.hip).  The seven sigma points are pushed through the camera model in the
// reference's order (centre, +x,+y,+z, -x,-y,-z) and the weighted mean / covariance are summed in that order: with alpha = 0.1 the UT
// weights are -99 / +16.67, so summation order is what decides +-1 px radii and hence tile membership.
#pragma once
#include "gsx_device.hpp"

namespace gsx {

struct UtProjOut { float radius_x, radius_y; f2 im; float depth, c00, c01, c11, ood, compensation; };

// `q` is the glm-normalised quaternion (w, x, y, z).  Returns false when the Gaussian is culled (the caller writes radii = 0 only, as
// upstream); true: `o` holds radii (as floats), the 2-D mean, the depth, the blurred covariance, 1/det and the compensation.
// GLOBAL_SHUTTER = true: the caller guarantees cams.shutter == GLOBAL (the fused front end): world_to_image is then its first projection alone —
// the same instructions on that path, but the rolling-shutter iteration (ten slerps with acos / sin per sigma point, unrolled seven
// times) is not compiled into the kernel at all.
template <int KIND, bool GLOBAL_SHUTTER = false>
GSX_DEV bool ut_project(const Camera<KIND>& cam, const ShutterPoses& sp, const f3 mean, const f3 scale, const quat q, const bool has_opacity,
                        const float opacity_in, const uint32_t W, const uint32_t H, const float eps2d, const float near_plane, const float far_plane,
                        const float radius_clip, const gsx_ut_params& ut, UtProjOut& o) {
    // camera-space depth at the centre-of-shutter pose (ProjectionUT3DGSFused.cu:74-82)
    f3 tc; quat qc;
    sp.at(0.5f, tc, qc);
    const f3 mean_c = quat_rotate(qc, mean) + tc;
    if (mean_c.z < near_plane || mean_c.z > far_plane) return false;

    // sigma points and weights
    const float D = 3.f;
    const float lambda = ut.alpha * ut.alpha * (D + ut.kappa) - D;
    const m33 R = quat_to_mat_raw(q);
    const float sq = sqrtf(D + lambda);
    const float w_m0 = lambda / (D + lambda);
    const float w_c0 = lambda / (D + lambda) + (1.f - ut.alpha * ut.alpha + ut.beta);
    const float w_i = 1.f / (2.f * (D + lambda));
    const float sc[3] = {scale.x, scale.y, scale.z};

    const bool require_all = ut.require_all_sigma_points_valid != 0;
    bool valid = require_all;
    f2 ipts[7];
    f2 im{0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        f3 pt = mean;
        if (i > 0) {
            const int ax = (i - 1) % 3;
            const float f = sq * sc[ax];
            const f3 delta{f * R.a[0][ax], f * R.a[1][ax], f * R.a[2][ax]};
            pt = (i <= 3) ? (mean + delta) : (mean - delta);
        }
        f2 ip;
        const bool pv = GLOBAL_SHUTTER ? cam.project(quat_rotate(sp.q0, pt) + sp.t0, ut.in_image_margin_factor, ip)
                                       : cam.world_to_image(pt, sp, ut.in_image_margin_factor, ip);
        if (require_all) {
            if (!pv) return false;
        } else {
            valid |= pv;
        }
        ipts[i] = ip;
        const float w = (i == 0) ? w_m0 : w_i;
        im.x += w * ip.x;
        im.y += w * ip.y;
    }
    if (!valid) return false;

    float c00 = 0.f, c01 = 0.f, c11 = 0.f;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const float dx = ipts[i].x - im.x, dy = ipts[i].y - im.y;
        const float w = (i == 0) ? w_c0 : w_i;
        c00 += w * (dx * dx);
        c01 += w * (dx * dy);
        c11 += w * (dy * dy);
    }
    // add_blur (Utils.cuh:171-179)
    const float det_orig = c00 * c11 - c01 * c01;
    c00 += eps2d; c11 += eps2d;
    const float det = c00 * c11 - c01 * c01;
    const float compensation = sqrtf(fmaxf(0.f, det_orig / det));
    if (det <= 0.f) return false;
    const float ood = 1.f / det;

    float extend = 3.33f;
    if (has_opacity) {
        float opacity = opacity_in;
        opacity *= compensation;
        if (opacity < (1.f / 255.f)) return false;
        extend = fminf(extend, sqrtf(2.f * __logf(opacity / (1.f / 255.f))));
    }
    const float b = 0.5f * (c00 + c11);
    const float tmp = sqrtf(fmaxf(0.01f, b * b - det));
    const float r1 = extend * sqrtf(b + tmp);
    const float radius_x = ceilf(fminf(extend * sqrtf(c00), r1));
    const float radius_y = ceilf(fminf(extend * sqrtf(c11), r1));
    if (radius_x <= radius_clip && radius_y <= radius_clip) return false;
    if (im.x + radius_x <= 0 || im.x - radius_x >= (float)W || im.y + radius_y <= 0 || im.y - radius_y >= (float)H) {
        return false;
    }
    o.radius_x = radius_x; o.radius_y = radius_y; o.im = im; o.depth = mean_c.z;
    o.c00 = c00; o.c01 = c01; o.c11 = c11; o.ood = ood; o.compensation = compensation;
    return true;
}

}  // namespace gsx
