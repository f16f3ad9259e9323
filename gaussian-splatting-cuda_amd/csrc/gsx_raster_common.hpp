// gsx_raster_common.hpp — declarations shared by the generic (gsx_raster.hip) and the fast
// (gsx_raster_fast.hip) world-space blend kernels.
#pragma once
#include "gsx_device.hpp"

namespace gsx {

void set_error(const char* msg);
int check_launch(const char* what);

constexpr int TILE = 16;
constexpr int RB = 256;  // threads per workgroup == Gaussians per chunk
constexpr float ALPHA_MIN = 1.f / 255.f;

struct RasterArgs {
    uint32_t C, N;
    int64_t n_isects;
    const float* means; const float* quats; const float* scales; const float* colors; const float* opacities;
    const float* backgrounds; const uint8_t* masks;
    uint32_t W, H, tw, th;
    gsx_cameras cams;
    const int32_t* tile_offsets; const int32_t* flatten_ids;
    const float4* packed;  // optional [C*N] x 64 B camera-space records (gsx_raster_fast.hip: pack_records_kernel), else nullptr
    // fisheye fast path only: per-(camera, tile) flag "some Gaussian of this tile's list has no usable (u0, v0) chart (it sits at or
    // beyond ~83 degrees off the optical axis)": the fast kernels skip flagged tiles, the generic kernels then run ONLY those
    const uint8_t* tile_flags;
};

constexpr size_t FAST_FLAG_BYTES = 262144;  // capacity of the tile-flag plane (C * tiles); larger grids take the generic kernels


// pixel owned by this thread: wave w owns the 8x8 quadrant (w&1, w>>1) of the tile, lane l the pixel (l&7, l>>3)
GSX_DEV void thread_pixel(uint32_t tid, uint32_t tile_x, uint32_t tile_y, uint32_t& i, uint32_t& j) {
    const uint32_t wave = tid >> 6, lane = tid & 63u;
    j = tile_x * TILE + (wave & 1u) * 8u + (lane & 7u);
    i = tile_y * TILE + (wave >> 1) * 8u + (lane >> 3);
}

// fast-path launchers (gsx_raster_fast.hip); kind is CAM_PERFECT_PINHOLE or CAM_OPENCV_PINHOLE, global shutter
// (fisheye: returns the tile-flag plane the caller hands to the generic kernel for the flagged tiles; nullptr otherwise)
const uint8_t* launch_raster_fwd_fast(int kind, RasterArgs a, float* renders, float* alphas, int32_t* last_ids, void* workspace,
                                      size_t workspace_bytes, hipStream_t st);
size_t raster_fwd_fast_workspace_bytes(uint32_t C, uint32_t N);
// returns false (nothing launched) when no sufficient workspace was supplied: the caller falls back to the generic kernels
bool launch_raster_bwd_fast(int kind, RasterArgs a, const float* render_alphas, const int32_t* last_ids,
                            const float* v_render_colors, const float* v_render_alphas, float* v_means, float* v_quats,
                            float* v_scales, float* v_colors, float* v_opacities, void* workspace, size_t workspace_bytes,
                            const float4* packed_from_fwd, hipStream_t st, const uint8_t** tile_flags_out);
size_t raster_bwd_fast_workspace_bytes(uint32_t C, uint32_t N, int64_t n_isects);

}  // namespace gsx
