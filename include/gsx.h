/* gsx.h — C ABI of libgsx.so: the MI355X (gfx950) implementation of the differentiable 3DGS
 * rasterizer hot path of MrNeRF/gaussian-splatting-cuda (`--gut` path, gsplat/Ops.h operator surface).
 *
 * This is the drop-in boundary: plain pointers (device memory unless stated), sizes, scalars and a
 * HIP stream.  No torch types, no allocation inside the library (callers pass outputs and, where
 * needed, a workspace sized by the matching *_workspace_bytes query).  Every entry point returns
 * GSX_OK (0) or a negative gsx_status; gsx_last_error() gives a thread-local message.
 * All launches are asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream).
 * The library is stateless and re-entrant, like the reference.
 *
 * Each function cites the reference interface it replaces (paths relative to /root/reference).
 * The C++ shim that re-creates `namespace gsplat` on at::Tensor over this ABI is
 * gaussian-splatting-cuda_amd/csrc/ops_shim.cpp; the binding a reference maintainer would add is
 * shown in INTEGRATION.md.
 */
#ifndef GSX_H
#define GSX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI history (gsx_abi_version() returns the value the library was built with; a host compares it with the header it compiled against):
 *   1  round 1.
 *   2  gsx_intersect_bin_fill gained the positional `max_segment` argument; the pinned host word written by gsx_intersect_bin_count holds
 *      n_isects in its low 32 bits (0xFFFFFFFF on overflow) and the largest tile segment in its high 32 bits; gsx_sh_colors_bwd accepts
 *      NULL radii / colors; ranked fill entry points added.
 *   3  additions only: gsx_frontend_fused(_supported), gsx_rasterize_to_pixels_from_world_3dgs_fwd_packed, gsx_intersect_tile_fill_packed.
 *   4  additions only: the guarded list protocol (no host read of n_isects on the render path): gsx_intersect_bin_count_guarded,
 *      gsx_rasterize_to_pixels_from_world_3dgs_{fwd,bwd}_guarded.
 *   5  gsx_intersect_bin_count(_guarded) store all ones into the pinned host word before they launch anything, and its high half is what the
 *      device writes last: the word can be POLLED by the host (no event in the stream); gsx_frontend_fused accepts NULL conics and takes `record_ranges`; gsx_splat_activations_bwd_reg added.
 *   6  gsx_rasterize_to_pixels_from_world_3dgs_bwd_guarded gained the positional `n_isects_expected` argument its forward twin already had: both
 *      choose their kernel variants from the same estimate (round 4's backward saw the capacity — 25 % above it — on guarded lists).
 *   7  additions only: gsx_rasterize_to_pixels_from_world_3dgs_bwd_act — the blend backward through to the raw SplatData parameters: the activation
 *      Jacobians as the epilogue of the gather kernel; gsx_photometric_loss_single_pass — the training loss and its gradient in one kernel. */
#define GSX_ABI_VERSION 7

typedef enum gsx_status {
    GSX_OK = 0,
    GSX_ERR_INVALID_ARGUMENT = -1, /* null pointer, bad size, unsupported channel count ... */
    GSX_ERR_UNSUPPORTED = -2,      /* e.g. ORTHO camera (rejected upstream too: Fwd.cu:128-132), packed mode */
    GSX_ERR_WORKSPACE_TOO_SMALL = -3,
    GSX_ERR_LAUNCH_FAILED = -4     /* hipGetLastError() != hipSuccess after a launch */
} gsx_status;

/* gsplat/Common.h:46-50 */
typedef enum gsx_camera_model { GSX_CAMERA_PINHOLE = 0, GSX_CAMERA_ORTHO = 1, GSX_CAMERA_FISHEYE = 2 } gsx_camera_model;

/* gsplat/Cameras.h:16-22 */
typedef enum gsx_shutter {
    GSX_SHUTTER_ROLLING_TOP_TO_BOTTOM = 0,
    GSX_SHUTTER_ROLLING_LEFT_TO_RIGHT = 1,
    GSX_SHUTTER_ROLLING_BOTTOM_TO_TOP = 2,
    GSX_SHUTTER_ROLLING_RIGHT_TO_LEFT = 3,
    GSX_SHUTTER_GLOBAL = 4
} gsx_shutter;

/* gsplat/Cameras.h:27-44 (UnscentedTransformParameters) */
typedef struct gsx_ut_params {
    float alpha;                             /* 0.1 */
    float beta;                              /* 2   */
    float kappa;                             /* 0   */
    float in_image_margin_factor;            /* 0.1 */
    int32_t require_all_sigma_points_valid;  /* 1   */
} gsx_ut_params;

/* The camera block shared by projection and rasterization (the `viewmats0 … thin_prism_coeffs`
 * argument run of gsplat/Ops.h:75-97, 113-125, 146-158).  All pointers are device pointers. */
typedef struct gsx_cameras {
    uint32_t C;               /* number of cameras */
    const float* viewmats0;   /* [C,4,4] row-major world->camera */
    const float* viewmats1;   /* [C,4,4] or NULL (rolling shutter end pose) */
    const float* Ks;          /* [C,3,3] */
    int32_t camera_model;     /* gsx_camera_model */
    int32_t shutter;          /* gsx_shutter */
    const float* radial;      /* [C,6] (pinhole) / [C,4] (fisheye) or NULL */
    const float* tangential;  /* [C,2] or NULL */
    const float* thin_prism;  /* [C,4] or NULL */
} gsx_cameras;

const char* gsx_last_error(void);
int gsx_abi_version(void);
/* Test / A-B switches (environment variables GSX_RASTER_PATH, GSX_BWD, GSX_INTERSECT, GSX_INTERSECT_FILL, GSX_BIN_NB) are honoured only when
 * GSX_TEST_SWITCHES=1 is set in the environment (read once per process): an embedding host inherits no hidden switch.  Returns the value
 * of `name` under that gate, else NULL. */
const char* gsx_test_switch(const char* name);

/* ---- spherical harmonics: gsplat/Ops.h:12-25, SphericalHarmonics.cpp:15-75 ------------------ */
/* dirs [n,3], coeffs [n,K,3], masks [n] bool(uint8) or NULL -> colors [n,3].
 * Masked-out elements are left untouched (as upstream, SphericalHarmonicsCUDA.cu:390-392). */
int gsx_spherical_harmonics_fwd(uint32_t degrees_to_use, uint32_t n, uint32_t K, const float* dirs,
                                const float* coeffs, const uint8_t* masks, float* colors, void* stream);
/* v_coeffs [n,K,3] is fully written (zeros for bases above the active degree and for masked
 * elements: replaces upstream's at::zeros_like + partial write); v_dirs [n,3] or NULL, fully written. */
int gsx_spherical_harmonics_bwd(uint32_t K, uint32_t degrees_to_use, uint32_t n, const float* dirs,
                                const float* coeffs, const uint8_t* masks, const float* v_colors, float* v_coeffs,
                                float* v_dirs, void* stream);

/* ---- projection: gsplat/Ops.h:69-98, Projection.cpp:22-110, ProjectionUT3DGSFused.cu ------- */
/* means [N,3], quats [N,4] wxyz, scales [N,3], opacities [N] or NULL ->
 * radii int32 [C,N,2], means2d [C,N,2], depths [C,N], conics [C,N,3], compensations [C,N] or NULL.
 * As upstream, only radii is written for culled Gaussians. */
int gsx_projection_ut_3dgs_fused(uint32_t N, const float* means, const float* quats, const float* scales,
                                 const float* opacities, const gsx_cameras* cams, uint32_t image_width,
                                 uint32_t image_height, float eps2d, float near_plane, float far_plane,
                                 float radius_clip, const gsx_ut_params* ut, int32_t* radii, float* means2d,
                                 float* depths, float* conics, float* compensations, void* stream);

/* ---- tile intersection: gsplat/Ops.h:28-43, Intersect.cpp:15-137, IntersectTile.cu --------- */
/* Phase 1 (Intersect.cpp:54-76): tiles_per_gauss int32 [C*N], cum_tiles_per_gauss int64 [C*N]
 * (inclusive scan), *n_isects_dev (device int64) = total.  If n_isects_host_pinned != NULL the total is
 * also copied there asynchronously (caller synchronises the stream before reading it). */
size_t gsx_intersect_count_workspace_bytes(uint32_t C, uint32_t N);
int gsx_intersect_tile_count(uint32_t C, uint32_t N, const float* means2d, const int32_t* radii, uint32_t tile_size,
                             uint32_t tile_width, uint32_t tile_height, int32_t* tiles_per_gauss,
                             int64_t* cum_tiles_per_gauss, int64_t* n_isects_dev, int64_t* n_isects_host_pinned,
                             void* workspace, size_t workspace_bytes, void* stream);
/* Phase 2 (Intersect.cpp:79-118, IntersectTile.cu:95-111,290-342): emit (key,value) pairs and
 * stable-sort them by the low 32+tile_n_bits+cam_n_bits key bits.  Outputs isect_ids int64 [n_isects],
 * flatten_ids int32 [n_isects]. */
size_t gsx_intersect_fill_workspace_bytes(uint32_t C, uint32_t N, int64_t n_isects, int sort);
int gsx_intersect_tile_fill(uint32_t C, uint32_t N, const float* means2d, const int32_t* radii, const float* depths,
                            const int64_t* cum_tiles_per_gauss, uint32_t tile_size, uint32_t tile_width,
                            uint32_t tile_height, int sort, int64_t n_isects, int64_t* isect_ids,
                            int32_t* flatten_ids, void* workspace, size_t workspace_bytes, void* stream);
/* The packed layout of the reference (Intersect.cpp:31-38; IntersectTile.cu:85-88): means2d [nnz,2], radii [nnz,2], depths [nnz] hold nnz
 * (camera, Gaussian) pairs, camera_ids int64 [nnz] names each pair's camera, the emitted flatten_ids index the nnz pairs.  Phase 1 is
 * gsx_intersect_tile_count(1, nnz, ...) (the count does not look at the camera); `N` is ignored. */
int gsx_intersect_tile_fill_packed(uint32_t C, uint32_t N, uint32_t nnz, const int64_t* camera_ids, const float* means2d,
                                   const int32_t* radii, const float* depths, const int64_t* cum_tiles_per_gauss,
                                   uint32_t tile_size, uint32_t tile_width, uint32_t tile_height, int sort, int64_t n_isects,
                                   int64_t* isect_ids, int32_t* flatten_ids, void* workspace, size_t workspace_bytes, void* stream);
/* Intersect.cpp:124-137: offsets int32 [C,tile_height,tile_width]. */
int gsx_intersect_offset(int64_t n_isects, const int64_t* isect_ids, uint32_t C, uint32_t tile_width,
                         uint32_t tile_height, int32_t* offsets, void* stream);

/* ---- world-space blend: gsplat/Ops.h:100-166, Rasterization.cpp:20-261 ---------------------- */
/* colors [C,N,3], opacities [C,N], backgrounds [C,3] or NULL, masks [C,th,tw] bool or NULL,
 * tile_offsets int32 [C,th,tw], flatten_ids int32 [n_isects] ->
 * renders [C,H,W,3], alphas [C,H,W,1], last_ids int32 [C,H,W].  channels must be 3. */
int gsx_rasterize_to_pixels_from_world_3dgs_fwd(uint32_t N, int64_t n_isects, const float* means,
                                                const float* quats, const float* scales, const float* colors,
                                                uint32_t channels, const float* opacities, const float* backgrounds,
                                                const uint8_t* masks, uint32_t image_width, uint32_t image_height,
                                                uint32_t tile_size, const gsx_cameras* cams, const gsx_ut_params* ut,
                                                const int32_t* tile_offsets, const int32_t* flatten_ids,
                                                float* renders, float* alphas, int32_t* last_ids, void* workspace,
                                                size_t workspace_bytes, void* stream);
/* Extension: `tile_size` = 32 means the LISTS (tile_offsets [C, ceil(H/32), ceil(W/32)], flatten_ids) were built for 32 x 32 pixel tiles
 * (intersect_tile / gsx_intersect_bin_* with tile_size 32): the kernels still work on 16 x 16 pixel tiles, each walking the list of its
 * 32 x 32 parent — the footprint tests drop what does not reach it, the image is the same.  For frames whose Gaussians cover many tiles
 * (a trained dense scene: 59 tiles per Gaussian) the intersection then handles 3x fewer keys.  Fast path only (global-shutter pinhole,
 * workspace given); `last_ids` index the coarse list; the backward needs gsx_rasterize_bwd_workspace_bytes(C, N, 4 * n_isects). */
/* `workspace` (optional; gsx_rasterize_fwd_workspace_bytes): room for one packed 64 B camera-space record per
 * (camera, Gaussian), so that staging a tile gathers ONE cache line per Gaussian instead of five (means, quats,
 * scales, opacities, colours live in five arrays).  NULL / too small = the reference-order (generic) kernels. */
size_t gsx_rasterize_fwd_workspace_bytes(uint32_t C, uint32_t N);
/* Gradient outputs v_means [N,3], v_quats [N,4], v_scales [N,3], v_colors [C,N,3], v_opacities [C,N] are OVERWRITTEN
 * (upstream accumulates into tensors its wrapper zero-fills, Rasterization.cpp:190-194: same values).  v_render_alphas
 * may be NULL (no gradient through the alpha output).
 * `workspace` (optional; size from gsx_rasterize_bwd_workspace_bytes): with it the fast path writes one 64 B
 * moment record per (tile, Gaussian), and a second kernel sums them and applies the chain rule once per
 * (camera, Gaussian), instead of issuing 14 device-scope float atomics per (tile, Gaussian) — on MI355X those
 * atomics are served memory-side (the 8 XCD L2s are not coherent) and cost more than the rest of the kernel.
 * NULL / too small = the reference-order kernels with float atomics. */
size_t gsx_rasterize_bwd_workspace_bytes(uint32_t C, uint32_t N, int64_t n_isects);
int gsx_rasterize_to_pixels_from_world_3dgs_bwd(uint32_t N, int64_t n_isects, const float* means,
                                                const float* quats, const float* scales, const float* colors,
                                                uint32_t channels, const float* opacities, const float* backgrounds,
                                                const uint8_t* masks, uint32_t image_width, uint32_t image_height,
                                                uint32_t tile_size, const gsx_cameras* cams, const gsx_ut_params* ut,
                                                const int32_t* tile_offsets, const int32_t* flatten_ids,
                                                const float* render_alphas, const int32_t* last_ids,
                                                const float* v_render_colors, const float* v_render_alphas,
                                                float* v_means, float* v_quats, float* v_scales, float* v_colors,
                                                float* v_opacities, void* workspace, size_t workspace_bytes,
                                                void* stream);
/* Fused per-Gaussian front end of one render (one global-shutter pinhole camera): SplatData activations -> UT projection -> SH colours
 * (+0.5, clamp_min 0) -> the packed 64 B records of the blend kernels, in ONE streaming kernel (csrc/gsx_frontend.hip: 376 B of HBM
 * traffic per visible Gaussian at SH degree 3 instead of ~470 B over four launches).  Outputs bit-identical to gsx_splat_activations_projection_ut
 * + gsx_sh_colors_fwd; records equal to rounding to the ones the blend forward packs for itself.  `fwd_workspace` (gsx_rasterize_fwd_workspace_bytes(1, N)) is then
 * handed to gsx_rasterize_to_pixels_from_world_3dgs_fwd_packed(..., records_ready = 1) and later to ..._bwd_packed.  Outputs: scales
 * [N,3], quats [N,4], opacities [N] (activated), radii int32 [1,N,2], means2d [1,N,2], depths [1,N], conics [1,N,3] or NULL (not
 * written: nothing on the render path reads them) (only radii for a culled Gaussian, as the projection), colors [1,N,3] (zero rows for culled Gaussians).  gsx_frontend_fused_supported: 1 when the
 * camera block / SH layout qualify (C == 1, PINHOLE with or without distortion, GLOBAL shutter, (K*3) % 4 == 0, 16 B aligned coeffs).
 * record_ranges = 1 (frames of large footprints: the caller's choice, e.g. whenever it builds lists per 32 x 32 pixels): the backward's per-(tile,
 * Gaussian) moment records of a Gaussian then occupy one contiguous run of slots sized by its rectangle of 16-pixel tiles, which the
 * gather streams; 0: they are chained per Gaussian (csrc/gsx_raster_common.hpp).  Same gradients either way. */
int gsx_frontend_fused_supported(uint32_t K, uint32_t degrees_to_use, const gsx_cameras* cams, const float* coeffs);
int gsx_frontend_fused(uint32_t N, uint32_t K, uint32_t degrees_to_use, const float* means, const float* rotation_raw,
                       const float* scaling_raw, const float* opacity_raw, const float* coeffs, const gsx_cameras* cams,
                       uint32_t image_width, uint32_t image_height, float eps2d, float near_plane, float far_plane, float radius_clip,
                       const gsx_ut_params* ut, float* scales, float* quats, float* opacities, int32_t* radii, float* means2d,
                       float* depths, float* conics, float* colors, void* fwd_workspace, size_t workspace_bytes, int record_ranges, void* stream);
/* The blend forward for a caller whose workspace already holds the packed records of exactly these inputs (records_ready = 1: written
 * by gsx_frontend_fused; 0 = gsx_rasterize_to_pixels_from_world_3dgs_fwd). */
int gsx_rasterize_to_pixels_from_world_3dgs_fwd_packed(uint32_t N, int64_t n_isects, const float* means,
                                                       const float* quats, const float* scales, const float* colors,
                                                       uint32_t channels, const float* opacities, const float* backgrounds,
                                                       const uint8_t* masks, uint32_t image_width, uint32_t image_height,
                                                       uint32_t tile_size, const gsx_cameras* cams, const gsx_ut_params* ut,
                                                       const int32_t* tile_offsets, const int32_t* flatten_ids,
                                                       float* renders, float* alphas, int32_t* last_ids, void* workspace,
                                                       size_t workspace_bytes, int records_ready, void* stream);
/* Same, for a caller that kept the forward's workspace alive: `packed_records` = gsx_rasterize_fwd_packed_records(the
 * workspace the forward of the SAME inputs ran with) lets the backward skip re-packing the per-(camera, Gaussian) records
 * (NULL = pack again).  The caller guarantees that neither the inputs nor that workspace changed in between. */
const void* gsx_rasterize_fwd_packed_records(const void* fwd_workspace, size_t workspace_bytes, uint32_t C, uint32_t N);
int gsx_rasterize_to_pixels_from_world_3dgs_bwd_packed(uint32_t N, int64_t n_isects, const float* means,
                                                       const float* quats, const float* scales, const float* colors,
                                                       uint32_t channels, const float* opacities, const float* backgrounds,
                                                       const uint8_t* masks, uint32_t image_width, uint32_t image_height,
                                                       uint32_t tile_size, const gsx_cameras* cams, const gsx_ut_params* ut,
                                                       const int32_t* tile_offsets, const int32_t* flatten_ids,
                                                       const float* render_alphas, const int32_t* last_ids,
                                                       const float* v_render_colors, const float* v_render_alphas,
                                                       float* v_means, float* v_quats, float* v_scales, float* v_colors,
                                                       float* v_opacities, void* workspace, size_t workspace_bytes,
                                                       const void* packed_records, void* stream);

/* ---- the remaining gsplat/Ops.h functions (Ops.h:45-65), so libgsx can stand in for the whole gsplat_backend --- */
/* gsplat::quats_to_rotmats, QuatToRotmatCUDA.cu:13-39: quats [N,4] wxyz -> rotmats [N,3,3] row-major */
int gsx_quats_to_rotmats(uint32_t N, const float* quats, float* rotmats, void* stream);
/* gsplat::relocation, Relocation.cpp:15-33 / RelocationCUDA.cu:11-43: ratios int32 [N], binoms [n_max,n_max] */
int gsx_relocation(uint32_t N, const float* opacities, const float* scales, const int32_t* ratios, const float* binoms,
                   int n_max, float* new_opacities, float* new_scales, void* stream);
/* gsplat::add_noise, Relocation.cpp:35-50 / RelocationCUDA.cu:112-141: means updated IN PLACE */
int gsx_add_noise(uint32_t N, const float* raw_opacities, const float* raw_scales, const float* raw_quats, const float* noise,
                  float* means, float current_lr, void* stream);

/* Binned variant of intersect_tile(sort = true) + intersect_offset in one pipeline (same outputs, bit for bit): per-block
 * LDS tile histograms -> per-tile prefix + exclusive scan = isect_offsets -> scatter into tile-major segments -> per-tile LDS
 * sort by (depth bits, flatten index).  No device-wide sort, no global atomics.  tile_offsets has C*tiles + 1 entries (the
 * last one is n_isects; the pinned host word receives n_isects in its low 32 bits — 0xFFFFFFFF when there are more than 2^31 - 1 —
 * and the key count of the largest (camera, tile) segment in its high 32 bits: segments up to 16384 keys are sorted in LDS by one
 * block or wave, larger ones as LDS-sorted 16384-key chunks plus ceil(log2(chunks)) merge-path passes by many blocks).
 * bin_fill's `max_segment` is an upper bound of that largest segment (it fixes the number of merge passes that are launched);
 * 0 = unknown (as many passes as n_isects could need: a few empty launches).  If a segment is larger than the stated bound its
 * part of flatten_ids / isect_ids is not written: re-run bin_fill with the value bin_count reported.
 * Host protocol as above: bin_count -> sync -> allocate
 * flatten_ids -> bin_fill(count_workspace = the workspace bin_count used).  isect_ids may be NULL (the blend kernels only
 * need flatten_ids + offsets).  bin_fill's `n_isects` is the CAPACITY of flatten_ids / isect_ids / the workspace: a caller
 * that can guess an upper bound may launch bin_fill before the host has read the exact total (nothing beyond the capacity is
 * written; if the total turns out larger, the outputs are incomplete and bin_fill must be re-run with enough room).  gsx_intersect_bin_supported: tile grids up to 36864 tiles per camera (LDS counters). */
int gsx_intersect_bin_supported(uint32_t tile_width, uint32_t tile_height);
size_t gsx_intersect_bin_count_workspace_bytes(uint32_t C, uint32_t tile_width, uint32_t tile_height);
int gsx_intersect_bin_count(uint32_t C, uint32_t N, const float* means2d, const int32_t* radii, uint32_t tile_size,
                            uint32_t tile_width, uint32_t tile_height, int32_t* tiles_per_gauss, int32_t* tile_offsets,
                            int64_t* n_isects_host_pinned, void* workspace, size_t workspace_bytes, void* stream);
size_t gsx_intersect_bin_fill_workspace_bytes(uint32_t C, uint32_t tile_width, uint32_t tile_height, int64_t n_isects);
int gsx_intersect_bin_fill(uint32_t C, uint32_t N, const float* means2d, const int32_t* radii, const float* depths,
                           uint32_t tile_size, uint32_t tile_width, uint32_t tile_height, const int32_t* tile_offsets,
                           int64_t n_isects, int64_t max_segment, const void* count_workspace, int32_t* flatten_ids, int64_t* isect_ids,
                           void* workspace, size_t workspace_bytes, void* stream);

/* Guarded lists: the render path WITHOUT a host read of n_isects.  A caller that has an upper-bound guess (`capacity` slots in flatten_ids,
 * merge passes for segments up to `max_segment` keys; 0 = no segment bound, as for the ranked fill) passes both to the count; bin_scan then
 * writes the verdict on the device: *lists_status (int32) = n_isects when the fill launched with these bounds produces complete lists, -1
 * when it does not.  The blend entry points below take that word: the last list ends at the total instead of at the host's `n_isects` (which is
 * then only the capacity of flatten_ids), and a frame whose verdict is -1 is rendered with EMPTY lists (background, zero gradients: no
 * unwritten slot of flatten_ids is ever read).  The host reads the same verdict whenever it likes — the pinned word of bin_count still
 * arrives: it holds all ones until the device has written it (high half last), so the host may poll it (host-coherent pinned memory:
 * hipHostMallocCoherent) instead of recording an event behind the count, which would put a system-scope release between the count and
 * the key scatter — and renders an overflowed frame again with enough room BEFORE it applies anything irreversible (an optimizer step).
 * Same results as the exact protocol whenever the verdict is >= 0; nothing blocks between the count and the blend. */
int gsx_intersect_bin_count_guarded(uint32_t C, uint32_t N, const float* means2d, const int32_t* radii, uint32_t tile_size,
                                    uint32_t tile_width, uint32_t tile_height, int32_t* tiles_per_gauss, int32_t* tile_offsets,
                                    int64_t* n_isects_host_pinned, void* workspace, size_t workspace_bytes, int64_t capacity,
                                    int64_t max_segment, int32_t* lists_status, void* stream);
/* gsx_rasterize_to_pixels_from_world_3dgs_{fwd,bwd}_packed on guarded lists: `n_isects` = capacity of flatten_ids (the backward's workspace
 * is sized by it), `lists_status` = the device word above (NULL = `n_isects` is exact: identical to the _packed entry points),
 * `n_isects_expected` = the caller's estimate of the total for launch decisions (which forward kernel, how many record chains per Gaussian in
 * the backward; 0 = use the capacity). */
int gsx_rasterize_to_pixels_from_world_3dgs_fwd_guarded(uint32_t N, int64_t n_isects, const float* means,
                                                        const float* quats, const float* scales, const float* colors,
                                                        uint32_t channels, const float* opacities, const float* backgrounds,
                                                        const uint8_t* masks, uint32_t image_width, uint32_t image_height,
                                                        uint32_t tile_size, const gsx_cameras* cams, const gsx_ut_params* ut,
                                                        const int32_t* tile_offsets, const int32_t* flatten_ids,
                                                        float* renders, float* alphas, int32_t* last_ids, void* workspace,
                                                        size_t workspace_bytes, int records_ready, const int32_t* lists_status,
                                                        int64_t n_isects_expected, void* stream);
int gsx_rasterize_to_pixels_from_world_3dgs_bwd_guarded(uint32_t N, int64_t n_isects, const float* means,
                                                        const float* quats, const float* scales, const float* colors,
                                                        uint32_t channels, const float* opacities, const float* backgrounds,
                                                        const uint8_t* masks, uint32_t image_width, uint32_t image_height,
                                                        uint32_t tile_size, const gsx_cameras* cams, const gsx_ut_params* ut,
                                                        const int32_t* tile_offsets, const int32_t* flatten_ids,
                                                        const float* render_alphas, const int32_t* last_ids,
                                                        const float* v_render_colors, const float* v_render_alphas,
                                                        float* v_means, float* v_quats, float* v_scales, float* v_colors,
                                                        float* v_opacities, void* workspace, size_t workspace_bytes,
                                                        const void* packed_records, const int32_t* lists_status,
                                                        int64_t n_isects_expected, void* stream);

/* ABI 7.  The guarded blend backward of ONE camera through to the RAW SplatData parameters (the reference applies these Jacobians with torch
 * autograd over splat_data.cpp:267-286: scales = exp(scaling_raw), quats = normalize(rotation_raw), opacities = sigmoid(opacity_raw)).
 * means / quats / scales / opacities must be those activations of the raw tensors.  On the fast path (global-shutter pinholes) the gather
 * kernel applies the Jacobians where it holds v_quats / v_scales / v_opacities in registers: the three are then NOT written (scratch) and
 * no gsx_splat_activations_bwd launch follows; elsewhere that kernel runs behind the blend backward.  v_scaling_raw [N,3], v_rotation_raw
 * [N,4], v_opacity_raw [N] are overwritten either way, with the values of gsx_splat_activations_bwd_reg (regulariser terms included). */
int gsx_rasterize_to_pixels_from_world_3dgs_bwd_act(uint32_t N, int64_t n_isects, const float* means,
                                                    const float* quats, const float* scales, const float* colors,
                                                    uint32_t channels, const float* opacities, const float* backgrounds,
                                                    const uint8_t* masks, uint32_t image_width, uint32_t image_height,
                                                    uint32_t tile_size, const gsx_cameras* cams, const gsx_ut_params* ut,
                                                    const int32_t* tile_offsets, const int32_t* flatten_ids,
                                                    const float* render_alphas, const int32_t* last_ids,
                                                    const float* v_render_colors, const float* v_render_alphas,
                                                    float* v_means, float* v_quats, float* v_scales, float* v_colors,
                                                    float* v_opacities, void* workspace, size_t workspace_bytes,
                                                    const void* packed_records, const int32_t* lists_status,
                                                    int64_t n_isects_expected, const float* scaling_raw, const float* rotation_raw,
                                                    const float* opacity_raw, float* v_scaling_raw, float* v_rotation_raw,
                                                    float* v_opacity_raw, float scale_reg_per_element, float opacity_reg_per_element,
                                                    void* stream);

/* Ranked variant of the fill for frames with heavy tiles (same outputs, bit for bit; same reference interface).  The Gaussians of
 * the frame are ranked once by (depth bits, flatten index) — gsx_intersect_depth_ranks: ranks[c*N + n] = position in that order,
 * order[rank] = flatten index; culled Gaussians rank last — and the per-tile keys are these 4-byte ranks.  A tile above 4096 keys
 * is sorted by setting its ranks in a C*N-bit bitmap in LDS and reading it back in order (no merge passes, any segment size);
 * lighter tiles use the same LDS merge sorts as bin_fill with half the bytes.  Pays off when most intersections sit in tiles above
 * 4096 keys (a trained dense scene); needs C*N <= 1 048 576 (128 KB of bitmap): gsx_intersect_ranked_supported.
 * Protocol: bin_count -> depth_ranks (any time before) -> bin_fill_ranked(count_workspace, ranks, order); `n_isects` is the
 * capacity of the outputs exactly as in bin_fill. */
int gsx_intersect_ranked_supported(uint32_t C, uint32_t N);
size_t gsx_intersect_depth_ranks_workspace_bytes(uint32_t C, uint32_t N);
int gsx_intersect_depth_ranks(uint32_t C, uint32_t N, const int32_t* radii, const float* depths, uint32_t* ranks, uint32_t* order,
                              void* workspace, size_t workspace_bytes, void* stream);
size_t gsx_intersect_bin_fill_ranked_workspace_bytes(int64_t n_isects);
int gsx_intersect_bin_fill_ranked(uint32_t C, uint32_t N, const float* means2d, const int32_t* radii, const float* depths,
                                  uint32_t tile_size, uint32_t tile_width, uint32_t tile_height, const int32_t* tile_offsets,
                                  int64_t n_isects, const void* count_workspace, const uint32_t* ranks, const uint32_t* order,
                                  int32_t* flatten_ids, int64_t* isect_ids, void* workspace, size_t workspace_bytes, void* stream);

/* ---- next tier: fused Adam step -------------------------------------------------------
 * fast_gs::optimizer::adam_step_wrapper, fastgs/optimizer/include/adam_kernels.cuh:13-38:
 *   m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= lr * bc1_rcp * m / (sqrt(v) * bc2_sqrt_rcp + eps)
 * param / grad are [rows, cols] with leading dimensions (elements) ld_param / ld_grad, the states contiguous. */
int gsx_adam_step(uint64_t rows, uint32_t cols, uint64_t ld_param, uint64_t ld_grad, float* param, float* exp_avg,
                  float* exp_avg_sq, const float* grad, float lr, float beta1, float beta2, float eps,
                  float bias_correction1_rcp, float bias_correction2_sqrt_rcp, void* stream);
/* Same update on a dense [rows, cols] tensor (cols % 4 == 0) whose columns [0, split) and [split, cols) are two parameter
 * groups sharing betas / eps / step count but not the learning rate (sh0 / shN of one [N,K,3] SH tensor:
 * strategy_utils.cpp:36-37); step_a / step_b == 0 leaves that block and its state untouched (fused_adam.cpp:68-76). */
int gsx_adam_step_split(uint64_t rows, uint32_t cols, uint32_t split, float* param, float* exp_avg, float* exp_avg_sq,
                        const float* grad, float lr_a, float lr_b, int step_a, int step_b, float beta1, float beta2, float eps,
                        float bias_correction1_rcp, float bias_correction2_sqrt_rcp, void* stream);

/* The same update on up to GSX_ADAM_MULTI_MAX dense tensors in ONE launch, each with its own learning rate and bias corrections (the
 * parameter groups of one optimizer step: fused_adam.cpp:20-96 loops over them with one launch each).  Arrays are host arrays. */
#define GSX_ADAM_MULTI_MAX 8
int gsx_adam_step_multi(uint32_t count, float* const* param, float* const* exp_avg, float* const* exp_avg_sq, const float* const* grad,
                        const uint64_t* n, const float* lr, const float* bias_correction1_rcp, const float* bias_correction2_sqrt_rcp,
                        float beta1, float beta2, float eps, void* stream);

/* ---- next tier: photometric loss ----------------------------------------------------------
 * fusedssim / fusedssim_backward, src/training/kernels/ssim.cu:436-470 / 478-510 (kernels :64-275, :283-428): planar
 * [B,CH,H,W] images, 11x11 Gaussian window (sigma 1.5), zero padding.  Pass dm_* = NULL for train == false. */
int gsx_fused_ssim_fwd(uint32_t B, uint32_t CH, uint32_t H, uint32_t W, float C1, float C2, const float* img1, const float* img2,
                       float* ssim_map, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12, void* stream);
int gsx_fused_ssim_bwd(uint32_t B, uint32_t CH, uint32_t H, uint32_t W, float C1, float C2, const float* img1, const float* img2,
                       const float* dL_dmap, float* dL_dimg1, const float* dm_dmu1, const float* dm_dsigma1_sq,
                       const float* dm_dsigma12, void* stream);
/* Trainer::compute_photometric_loss (src/training/trainer.cpp:103-127) fused with the image clamp / permute of
 * rasterizer.cpp:401: loss = (1-lambda) * L1(clamp(render), gt) + lambda * (1 - mean(SSIM "valid" map)).
 * render / v_render: the blend's [C,H,W,3] layout; gt: [C,3,H,W]; loss3 = {loss, l1, ssim} on the device.
 * bwd multiplies by grad_scale and, when non-NULL, by the device scalar *grad_loss.  The same workspace must be passed
 * to bwd after fwd. */
size_t gsx_photometric_loss_workspace_bytes(uint32_t C, uint32_t H, uint32_t W);
int gsx_photometric_loss_fwd(uint32_t C, uint32_t H, uint32_t W, float lambda_dssim, const float* render, const float* gt,
                             float* loss3, void* workspace, size_t workspace_bytes, void* stream);
int gsx_photometric_loss_bwd(uint32_t C, uint32_t H, uint32_t W, float lambda_dssim, const float* grad_loss, float grad_scale,
                             const float* render, const float* gt, const void* workspace, size_t workspace_bytes, float* v_render,
                             void* stream);
/* ABI 7: the training loss and its gradient in ONE kernel (the SSIM statistics recomputed on a 5-pixel ring, the derivative maps in LDS only):
 * loss3 as gsx_photometric_loss_fwd, v_render [C,H,W,3] = grad_scale * d loss / d render as gsx_photometric_loss_bwd with grad_loss = NULL.
 * Same values as the pair; which of the two is faster is a measurement.  Workspace: the per-tile partial sums only. */
size_t gsx_photometric_loss_single_pass_workspace_bytes(uint32_t C, uint32_t H, uint32_t W);
int gsx_photometric_loss_single_pass(uint32_t C, uint32_t H, uint32_t W, float lambda_dssim, float grad_scale, const float* render,
                                     const float* gt, float* loss3, float* v_render, void* workspace, size_t workspace_bytes, void* stream);

/* ---- fused glue (extensions beyond gsplat/Ops.h) --------------------------------------------------
 * The reference's render glue wraps the operators in chains of small torch ops every frame; on MI355X those
 * ~40 launches cost as much as a blend kernel.  These entry points fuse them; results are identical to the
 * unfused sequence (tests/test_gpu_fused.py).
 *   gsx_sh_colors_*:        rasterizer.cpp:250-266  campos/dirs/masks/SH/+0.5/clamp_min (and their backward,
 *                           incl. the dirs->means gradient and the sum over cameras of the broadcast coeffs)
 *   gsx_splat_activations_*: splat_data.cpp:267-286  exp / normalize / sigmoid (and their backward) */
int gsx_sh_colors_fwd(uint32_t degrees_to_use, uint32_t C, uint32_t N, uint32_t K, const float* means,
                      const float* viewmats, const float* coeffs, const int32_t* radii, float* colors, void* stream);
/* v_coeffs [N,K,3] fully written; v_means_out [N,3] = (v_means_in or 0) + d colors / d means.
 * radii == colors == NULL: v_colors [C,N,3] are "pre-masked" (each camera's rows already carry its visibility and clamp masks, zero
 * where the camera does not see the Gaussian): the mode of the multi-GPU colour-gradient exchange, where the C cameras of a step were
 * rendered on C ranks and only these 3 floats per (camera, Gaussian) travel instead of the K*3 SH gradients. */
int gsx_sh_colors_bwd(uint32_t degrees_to_use, uint32_t C, uint32_t N, uint32_t K, const float* means,
                      const float* viewmats, const float* coeffs, const int32_t* radii, const float* colors,
                      const float* v_colors, float* v_coeffs, const float* v_means_in, float* v_means_out, void* stream);
/* gsx_sh_colors_bwd fused with the Adam step of the SH tensor (groups sh0 = first 3 floats of a row / shN = the rest of
 * src/training/optimizers/fused_adam.cpp:20-96; arithmetic of gsx_adam_step_split): coeffs, exp_avg, exp_avg_sq [N,K,3] are updated in
 * place and the SH gradient is never written (1 M Gaussians: 576 MB of HBM traffic less per training iteration).  K * 3 % 4 == 0.
 * step_* = lr * bias_correction1_rcp of the group; do_* = 0 leaves the group's block untouched (the shN warm-up quirk). */
int gsx_sh_colors_bwd_adam(uint32_t degrees_to_use, uint32_t C, uint32_t N, uint32_t K, const float* means,
                           const float* viewmats, float* coeffs, const int32_t* radii, const float* colors,
                           const float* v_colors, const float* v_means_in, float* v_means_out, float* exp_avg,
                           float* exp_avg_sq, float step_sh0, float step_shN, int do_sh0, int do_shN, float beta1, float beta2,
                           float eps, float bias_correction2_sqrt_rcp, void* stream);
/* gsx_splat_activations_fwd followed by gsx_projection_ut_3dgs_fused (one camera) in ONE launch: raw parameters in, the activated
 * copies (scales, quats, opacities) and the projection (radii, means2d, depths, conics) out; bit-identical to the two calls. */
int gsx_splat_activations_projection_ut(uint32_t N, const float* means, const float* rotation_raw, const float* scaling_raw,
                                        const float* opacity_raw, const gsx_cameras* cams, uint32_t image_width,
                                        uint32_t image_height, float eps2d, float near_plane, float far_plane, float radius_clip,
                                        const gsx_ut_params* ut, float* scales, float* quats, float* opacities, int32_t* radii,
                                        float* means2d, float* depths, float* conics, void* stream);
int gsx_splat_activations_fwd(uint32_t N, const float* scaling_raw, const float* rotation_raw, const float* opacity_raw,
                              float* scales, float* quats, float* opacities, void* stream);
int gsx_splat_activations_bwd(uint32_t N, const float* scaling_raw, const float* rotation_raw, const float* opacity_raw,
                              const float* v_scales, const float* v_quats, const float* v_opacities, float* v_scaling_raw,
                              float* v_rotation_raw, float* v_opacity_raw, void* stream);
/* The same with the gradients of the MCMC strategy's two regularisers added in place (scale_reg * mean(exp(scaling_raw)) and opacity_reg *
 * mean(sigmoid(opacity_raw)): /root/reference/src/training/trainer.cpp:103-127 adds them to the loss): pass reg / numel of the tensor; 0 = none. */
int gsx_splat_activations_bwd_reg(uint32_t N, const float* scaling_raw, const float* rotation_raw, const float* opacity_raw,
                                  const float* v_scales, const float* v_quats, const float* v_opacities, float* v_scaling_raw,
                                  float* v_rotation_raw, float* v_opacity_raw, float scale_reg_per_element, float opacity_reg_per_element,
                                  void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GSX_H */
