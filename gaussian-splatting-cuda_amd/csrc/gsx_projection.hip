// gsx_projection.hip — 3DGUT unscented-transform projection for gfx950 (non-differentiable).
//
// Replaces gsplat::projection_ut_3dgs_fused (reference: gsplat/Projection.cpp:22-110, kernel
// gsplat/ProjectionUT3DGSFused.cu:16-203, sigma points / UT gsplat/Cameras.cuh:1028-1150).
//
// Streaming op: 44 B in + 32 B out per (camera, Gaussian), ~500 flops.  One lane per Gaussian,
// grid.y = camera so every camera quantity is block-uniform.  The seven sigma points are pushed
// through the camera model in the reference's order (centre, +x,+y,+z, -x,-y,-z) and the weighted
// mean / covariance are summed in that order: with alpha = 0.1 the UT weights are -99 / +16.67, so
// summation order is what decides +-1 px radii and hence tile membership.
#include "gsx_device.hpp"
#include "gsx_ut_project.hpp"

namespace gsx {

void set_error(const char* msg);
int check_launch(const char* what);

constexpr int PROJ_BLOCK = 256;

// SplatData activations fused in front of the projection (gsx_splat_activations_projection_ut): the inputs are the RAW parameters
// (log-scales, un-normalised quaternions, opacity logits); the kernel writes the activated copies the later stages read and goes on
// with them — the same values, in the same order of operations, as gsx_splat_activations_fwd followed by the projection.
struct ProjActOut { float* scales; float* quats; float* opacities; };

template <int KIND, bool RAW>
__global__ __launch_bounds__(PROJ_BLOCK) void projection_ut_kernel(
    uint32_t N, const float* __restrict__ means, const float* __restrict__ quats, const float* __restrict__ scales,
    const float* __restrict__ opacities, gsx_cameras cams, uint32_t W, uint32_t H, float eps2d, float near_plane,
    float far_plane, float radius_clip, gsx_ut_params ut, int32_t* __restrict__ radii, float* __restrict__ means2d,
    float* __restrict__ depths, float* __restrict__ conics, float* __restrict__ compensations, ProjActOut act) {
    const uint32_t gid = blockIdx.x * PROJ_BLOCK + threadIdx.x;
    const uint32_t cid = blockIdx.y;
    if (gid >= N) return;
    const size_t idx = (size_t)cid * N + gid;

    const Camera<KIND> cam(cams, cid, W, H);
    const ShutterPoses sp(cams.viewmats0 + cid * 16, cams.viewmats1 ? cams.viewmats1 + cid * 16 : nullptr);

    const f3 mean{means[(size_t)gid * 3], means[(size_t)gid * 3 + 1], means[(size_t)gid * 3 + 2]};
    f3 scale{scales[(size_t)gid * 3], scales[(size_t)gid * 3 + 1], scales[(size_t)gid * 3 + 2]};
    quat q{quats[(size_t)gid * 4], quats[(size_t)gid * 4 + 1], quats[(size_t)gid * 4 + 2], quats[(size_t)gid * 4 + 3]};
    float opacity_in = opacities != nullptr ? opacities[gid] : 0.f;
    if (RAW) {   // splat_activations_fwd_kernel, verbatim (splat_data.cpp:267-286)
        scale = {expf(scale.x), expf(scale.y), expf(scale.z)};
        const float inv = 1.f / fmaxf(sqrtf(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z), 1e-12f);   // (quat fields hold the raw components in memory order)
        q = {q.w * inv, q.x * inv, q.y * inv, q.z * inv};
        opacity_in = 1.f / (1.f + expf(-opacity_in));
        act.scales[(size_t)gid * 3] = scale.x; act.scales[(size_t)gid * 3 + 1] = scale.y; act.scales[(size_t)gid * 3 + 2] = scale.z;
        reinterpret_cast<float4*>(act.quats)[gid] = make_float4(q.w, q.x, q.y, q.z);
        act.opacities[gid] = opacity_in;
    }
    {   // glm::normalize(quat)
        const float len = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
        if (len <= 0.f) q = {1.f, 0.f, 0.f, 0.f};
        else { const float o = 1.f / len; q = {q.w * o, q.x * o, q.y * o, q.z * o}; }
    }

    UtProjOut o;
    if (!ut_project<KIND>(cam, sp, mean, scale, q, opacities != nullptr, opacity_in, W, H, eps2d, near_plane, far_plane, radius_clip, ut, o)) {
        radii[idx * 2] = 0; radii[idx * 2 + 1] = 0;   // as upstream, only radii is written for a culled Gaussian
        return;
    }
    radii[idx * 2] = (int32_t)o.radius_x;
    radii[idx * 2 + 1] = (int32_t)o.radius_y;
    means2d[idx * 2] = o.im.x;
    means2d[idx * 2 + 1] = o.im.y;
    depths[idx] = o.depth;
    conics[idx * 3] = o.c11 * o.ood;
    conics[idx * 3 + 1] = -o.c01 * o.ood;
    conics[idx * 3 + 2] = o.c00 * o.ood;
    if (compensations != nullptr) compensations[idx] = o.compensation;
}

}  // namespace gsx

using namespace gsx;

static int launch_projection(bool raw, uint32_t N, const float* means, const float* quats, const float* scales, const float* opacities,
                             const gsx_cameras* cams, uint32_t image_width, uint32_t image_height, float eps2d, float near_plane,
                             float far_plane, float radius_clip, const gsx_ut_params* ut, int32_t* radii, float* means2d, float* depths,
                             float* conics, float* compensations, ProjActOut act, void* stream, const char* what) {
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((N + PROJ_BLOCK - 1) / PROJ_BLOCK, cams->C), block(PROJ_BLOCK);
#define GSX_LAUNCH_PROJ(KIND)                                                                                                       \
    do {                                                                                                                           \
        if (raw) hipLaunchKernelGGL(HIP_KERNEL_NAME(projection_ut_kernel<KIND, true>), grid, block, 0, st, N, means, quats, scales, \
                                    opacities, *cams, image_width, image_height, eps2d, near_plane, far_plane, radius_clip, *ut,    \
                                    radii, means2d, depths, conics, compensations, act);                                            \
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(projection_ut_kernel<KIND, false>), grid, block, 0, st, N, means, quats, scales,    \
                                opacities, *cams, image_width, image_height, eps2d, near_plane, far_plane, radius_clip, *ut, radii, \
                                means2d, depths, conics, compensations, act);                                                       \
    } while (0)
    if (cams->camera_model == GSX_CAMERA_PINHOLE) {
        if (!cams->radial && !cams->tangential && !cams->thin_prism) GSX_LAUNCH_PROJ(CAM_PERFECT_PINHOLE);
        else GSX_LAUNCH_PROJ(CAM_OPENCV_PINHOLE);
    } else if (cams->camera_model == GSX_CAMERA_FISHEYE) {
        GSX_LAUNCH_PROJ(CAM_OPENCV_FISHEYE);
    } else {
        set_error("projection_ut_3dgs_fused: unsupported camera model (only PINHOLE and FISHEYE; the reference asserts)");
        return GSX_ERR_UNSUPPORTED;
    }
#undef GSX_LAUNCH_PROJ
    return check_launch(what);
}

extern "C" int gsx_projection_ut_3dgs_fused(uint32_t N, const float* means, const float* quats, const float* scales,
                                            const float* opacities, const gsx_cameras* cams, uint32_t image_width,
                                            uint32_t image_height, float eps2d, float near_plane, float far_plane,
                                            float radius_clip, const gsx_ut_params* ut, int32_t* radii, float* means2d,
                                            float* depths, float* conics, float* compensations, void* stream) {
    if (!cams || !ut) { set_error("projection_ut_3dgs_fused: cams/ut is null"); return GSX_ERR_INVALID_ARGUMENT; }
    if (N == 0 || cams->C == 0) return GSX_OK;  // upstream skips the launch (ProjectionUT3DGSFused.cu:242-245)
    if (!means || !quats || !scales || !cams->viewmats0 || !cams->Ks || !radii || !means2d || !depths || !conics) {
        set_error("projection_ut_3dgs_fused: null pointer");
        return GSX_ERR_INVALID_ARGUMENT;
    }
    return launch_projection(false, N, means, quats, scales, opacities, cams, image_width, image_height, eps2d, near_plane, far_plane,
                             radius_clip, ut, radii, means2d, depths, conics, compensations, ProjActOut{nullptr, nullptr, nullptr}, stream,
                             "projection_ut_3dgs_fused");
}

// gsx_splat_activations_fwd + gsx_projection_ut_3dgs_fused in one launch (one camera): raw parameters in, activated copies AND the
// projection out.  Saves a launch and the re-read of the activated parameters (44 B per Gaussian); results are bit-identical.
extern "C" int gsx_splat_activations_projection_ut(uint32_t N, const float* means, const float* rotation_raw, const float* scaling_raw,
                                                   const float* opacity_raw, const gsx_cameras* cams, uint32_t image_width,
                                                   uint32_t image_height, float eps2d, float near_plane, float far_plane,
                                                   float radius_clip, const gsx_ut_params* ut, float* scales, float* quats,
                                                   float* opacities, int32_t* radii, float* means2d, float* depths, float* conics,
                                                   void* stream) {
    if (!cams || !ut) { set_error("splat_activations_projection_ut: cams/ut is null"); return GSX_ERR_INVALID_ARGUMENT; }
    if (cams->C != 1) { set_error("splat_activations_projection_ut: one camera per call (the activations do not depend on the camera)"); return GSX_ERR_UNSUPPORTED; }
    if (N == 0) return GSX_OK;
    if (!means || !rotation_raw || !scaling_raw || !opacity_raw || !cams->viewmats0 || !cams->Ks || !scales || !quats || !opacities || !radii ||
        !means2d || !depths || !conics) {
        set_error("splat_activations_projection_ut: null pointer");
        return GSX_ERR_INVALID_ARGUMENT;
    }
    return launch_projection(true, N, means, rotation_raw, scaling_raw, opacity_raw, cams, image_width, image_height, eps2d, near_plane,
                             far_plane, radius_clip, ut, radii, means2d, depths, conics, nullptr, ProjActOut{scales, quats, opacities}, stream,
                             "splat_activations_projection_ut");
}
