// gsx_raster_common.hpp — declarations shared by the generic (gsx_raster.hip) and the fast
// (gsx_raster_fast.hip) world-space blend kernels.
#pragma once
#include "gsx_device.hpp"

namespace gsx {

void set_error(const char* msg);
int check_launch(const char* what);
const char* test_switch(const char* name);   // gsx_capi.hip: getenv gated by GSX_TEST_SWITCHES=1

constexpr int TILE = 16;
constexpr int RB = 256;  // threads per workgroup == Gaussians per chunk
constexpr float ALPHA_MIN = 1.f / 255.f;

struct RasterArgs {
    uint32_t C, N;
    int64_t n_isects;
    const float* means; const float* quats; const float* scales; const float* colors; const float* opacities;
    const float* backgrounds; const uint8_t* masks;
    uint32_t W, H, tw, th;
    // The lists (tile_offsets / flatten_ids) may have been built for tiles of 16 << lshift pixels (lshift = 1: one list per 2 x 2 pixel
    // tiles, an extension of the fused path for frames whose Gaussians cover many tiles — the pixel tiles of the kernels stay 16 x 16, a
    // tile walks the list of its 32 x 32 parent and the footprint tests drop what does not reach it).  ltw x lth = the list grid.
    uint32_t lshift, ltw, lth;
    uint32_t rect_filter;   // lshift != 0: drop list entries whose rectangle of 16-px tiles excludes the tile (1; 0 only by the test switch GSX_LIST_RECT=0)
    gsx_cameras cams;
    const int32_t* tile_offsets; const int32_t* flatten_ids;
    const float4* packed;  // optional [C*N] x 64 B camera-space records (gsx_raster_fast.hip: pack_records_kernel), else nullptr
    // fisheye fast path only: per-(camera, tile) flag "some Gaussian of this tile's list has no usable (u0, v0) chart (it sits at or
    // beyond ~83 degrees off the optical axis)": the fast kernels skip flagged tiles, the generic kernels then run ONLY those
    const uint8_t* tile_flags;
    // Guarded lists (gsx_intersect_bin_count_guarded): `n_isects` is then only the CAPACITY of flatten_ids and *lists_status holds the
    // frame's true total — or -1 when the total (or the largest segment) outgrew what the optimistic fill was launched with.  The last
    // list ends at the total; an overflowed frame has EMPTY lists (background image, zero gradients: nothing unwritten is ever read)
    // and the host, which reads the same verdict later, renders it again.  nullptr = `n_isects` is exact (the reference's protocol).
    const int32_t* lists_status;
    int64_t n_isects_expected;   // list-density estimate for launch decisions (kernel variants); = n_isects when exact
    uint32_t chain_mask;         // backward: NSUB - 1 = NSUB record chains per (camera, Gaussian), 0 = one (see tile_chain)
    int64_t rec_capacity;        // backward: record slots behind ws_rec (ranges: a slot beyond it is never written)
};

// Optional epilogue of the backward's gather kernel (round 6): the SplatData activation Jacobians (splat_data.cpp:267-286: scales = exp(raw),
// quats = normalize(raw), opacities = sigmoid(raw)) applied where the gather holds v_quats / v_scales / v_opacities of a Gaussian in
// registers — the raw-parameter gradients leave directly, the three activated-parameter gradients are never written or re-read and the
// splat_activations_bwd launch (19 us, 124 MB at S-1M) disappears.  One camera only (C == 1: opacities are per camera).  Same arithmetic
// as splat_activations_bwd_kernel (gsx_sh.hip), on the ACTIVATED scales / opacities the blend was given (bit-identical to exp / sigmoid
// of the raw values: the front end computes them with the same expressions) and the RAW quaternion (its norm is not in the activated one).
struct ActEpilogue {
    const float* rotation_raw;   // [N,4]; nullptr = no epilogue
    float* v_scaling_raw; float* v_rotation_raw; float* v_opacity_raw;   // [N,3], [N,4], [N]
    float scale_reg, opacity_reg;   // regulariser gradients per element (gsx_splat_activations_bwd_reg), 0 = none
};

// end of the LAST list of the frame (every other list ends where the next one starts); `ok` = false: overflowed frame, all lists empty
GSX_DEV int32_t lists_total(const RasterArgs& a, bool& ok) {
    ok = true;
    if (a.lists_status == nullptr) return (int32_t)a.n_isects;
    const int32_t tot = *a.lists_status;   // wave-uniform address: one scalar load
    ok = tot >= 0;
    return tot;
}

__host__ __device__ inline size_t align256(size_t v) { return (v + 255) / 256 * 256; }
constexpr size_t FAST_FLAG_BYTES = 262144;  // capacity of the tile-flag plane (C * tiles); larger grids take the generic kernels


// range [start, end) of pixel tile (tile_x, tile_y) of camera cid in the sorted intersection list, and the slot of that tile's per-list-entry
// records (the backward's moment records): with lists per 32 x 32 pixels the four pixel tiles that share a list entry get four slots
GSX_DEV void tile_list_range(const RasterArgs& a, uint32_t cid, uint32_t tile_x, uint32_t tile_y, int32_t& start, int32_t& end) {
    const uint32_t lt = (tile_y >> a.lshift) * a.ltw + (tile_x >> a.lshift), n_lt = a.ltw * a.lth;
    const int32_t* toff = a.tile_offsets + (size_t)cid * n_lt;
    bool ok;
    const int32_t total = lists_total(a, ok);
    start = toff[lt];
    end = (cid == a.C - 1 && lt == n_lt - 1) ? total : toff[lt + 1];
    if (!ok) end = start;
}
// The backward chains the moment records of a (camera, Gaussian) for the gather kernel.  On frames of large footprints (a Gaussian
// covering w x h tiles has a chain of w h records) the walk — a pointer chase, one dependent 64 B load per record, the longest chain of a
// wave is what the wave takes — dominated the backward: such frames use NSUB chains per Gaussian, chosen by the parity of the pixel tile
// (four chains of ~w h / 4 records, walked side by side with four record loads in flight per thread).  Frames of small footprints keep
// ONE chain (a.chain_mask = 0): their gather is bound by the bytes it moves, and three more heads per Gaussian are just more bytes.
// Head array: NSUB planes of [C*N] int32, plane c = heads of chain c (-1 = empty); only the planes of the frame's chains are read.
//
// RANGES (round 4, mode word = 1: the fused front end packed the workspace of a frame of large footprints): the front end knows every
// Gaussian's rectangle of 16-pixel tiles, i.e. an upper bound of the records the backward can write for it, and gives every Gaussian that
// many CONSECUTIVE record slots: plane 0 = its offset inside its wave of 64 Gaussians (a DPP scan), the waves' totals are scanned by one
// small launch behind the front end into "first slot of wave w" (no atomics: 15 k returning adds on one allocator word cost 145 us).
// Plane 1 = records claimed so far (0): the backward claims slot first + (returning add on plane 1) — no chain link —, the gather reads
// [first, first + count) as one contiguous run (neighbouring Gaussians' runs are neighbours in memory: it streams instead of chasing
// 64 B pointers; a run longer than a few records is summed by the whole wave) and puts the count back to 0.  Chains remain the layout of
// frames of small footprints (nothing to gain there: S-1M's gather is bound by its bytes) and of workspaces packed by
// pack_records_kernel (the operator-level entry points: no rectangles there).
constexpr int NSUB = 4;
constexpr uint32_t REC_MODE_CHAINS = 0u, REC_MODE_RANGES = 1u;
GSX_DEV uint32_t tile_chain(const RasterArgs& a, uint32_t tile_x, uint32_t tile_y) { return ((tile_x & 1u) | ((tile_y & 1u) << 1)) & a.chain_mask; }
GSX_DEV int32_t tile_record_slot(const RasterArgs& a, int32_t isect, uint32_t tile_x, uint32_t tile_y) {
    return a.lshift ? (isect << 2) | (int32_t)(((tile_y & 1u) << 1) | (tile_x & 1u)) : isect;
}

// pixel owned by this thread: wave w owns the 8x8 quadrant (w&1, w>>1) of the tile, lane l the pixel (l&7, l>>3)
GSX_DEV void thread_pixel(uint32_t tid, uint32_t tile_x, uint32_t tile_y, uint32_t& i, uint32_t& j) {
    const uint32_t wave = tid >> 6, lane = tid & 63u;
    j = tile_x * TILE + (wave & 1u) * 8u + (lane & 7u);
    i = tile_y * TILE + (wave >> 1) * 8u + (lane >> 3);
}

// reduce x[0..15] over the 64 lanes; every lane of quad q = (lane >> 2) & 3 of row r = lane >> 4 returns the total of value
// 4*q + {0,2,1,3}[r].
// The swaps are issued through inline asm: with hipcc/ROCm 7.2 `r[0] + r[1]` on the result of
// __builtin_amdgcn_permlane{32,16}_swap compiles to `v_add v, vdst, vdst` (the second result is lost; see
// tools/butterfly_probe.hip).  `s_nop 1` = the two wait states a VALU write needs before v_permlane*_swap reads it.
GSX_DEV float butterfly_reduce16(float (&x)[16]) {
    asm volatile("s_nop 1\n\t"
                 "v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\t"
                 "v_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\t"
                 "v_permlane32_swap_b32 %8, %9\n\tv_permlane32_swap_b32 %10, %11\n\t"
                 "v_permlane32_swap_b32 %12, %13\n\tv_permlane32_swap_b32 %14, %15\n\t"
                 "s_nop 1"
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]),
                   "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]));
    float y[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) y[j] = x[2 * j] + x[2 * j + 1];
    asm volatile("s_nop 1\n\t"
                 "v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\t"
                 "v_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7\n\t"
                 "s_nop 1"
                 : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7]));
    // Inside a row the halving continues (4 values x 16 lanes -> 1 value per lane): exchange distance 8 (row_ror:8) keeps the
    // pair selected by lane bit 3, distance 4 (ds_swizzle xor 4, LDS crossbar: no VALU slot) the value selected by lane bit 2,
    // then the quad is summed with two quad_perm adds.  12 VALU instead of 20 for the four 16-lane row sums.
    const uint32_t ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const bool hi8 = (ln & 8u) != 0u, hi4 = (ln & 4u) != 0u;
    const float v0 = y[0] + y[1], v1 = y[2] + y[3], v2 = y[4] + y[5], v3 = y[6] + y[7];
    float k0 = hi8 ? v2 : v0, k1 = hi8 ? v3 : v1;
    const float s0 = hi8 ? v0 : v2, s1 = hi8 ? v1 : v3;
    k0 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s0), 0x128, 0xf, 0xf, true));  // row_ror:8
    k1 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s1), 0x128, 0xf, 0xf, true));
    float k = hi4 ? k1 : k0;
    const float sd = hi4 ? k0 : k1;
    k += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, sd), 0x101F));                   // lane ^ 4
    k += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, k), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
    k += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, k), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
    return k;
}

// moment whose wave total butterfly_reduce16 leaves in `lane` (valid in lanes with (lane & 3) == 0): 4 * quad + {0,2,1,3}[row]
GSX_DEV uint32_t butterfly_value_of_lane(uint32_t lane) {
    const uint32_t row = lane >> 4;
    return 4u * ((lane >> 2) & 3u) + ((row == 1u) ? 2u : (row == 2u ? 1u : row));
}

// fast-path launchers (gsx_raster_fast.hip); kind is CAM_PERFECT_PINHOLE or CAM_OPENCV_PINHOLE, global shutter
// (fisheye: returns the tile-flag plane the caller hands to the generic kernel for the flagged tiles; nullptr otherwise)
const uint8_t* launch_raster_fwd_fast(int kind, RasterArgs a, float* renders, float* alphas, int32_t* last_ids, void* workspace,
                                      size_t workspace_bytes, hipStream_t st, bool records_ready = false);
size_t raster_fwd_fast_workspace_bytes(uint32_t C, uint32_t N);
int32_t* raster_fwd_fast_heads(const float4* packed, uint32_t C, uint32_t N);
uint32_t* raster_fwd_fast_alloc(const float4* packed, uint32_t C, uint32_t N);   // word [1] = REC_MODE_* of the head planes; from word 64 on: first record slot of every wave of 64 Gaussians (ranges)   // the chain-head array [C*N][NSUB] inside the forward workspace (see its layout)
// returns false (nothing launched) when no sufficient workspace was supplied: the caller falls back to the generic kernels
bool launch_raster_bwd_fast(int kind, RasterArgs a, const float* render_alphas, const int32_t* last_ids,
                            const float* v_render_colors, const float* v_render_alphas, float* v_means, float* v_quats,
                            float* v_scales, float* v_colors, float* v_opacities, void* workspace, size_t workspace_bytes,
                            const float4* packed_from_fwd, hipStream_t st, const uint8_t** tile_flags_out, const ActEpilogue* act = nullptr);
size_t raster_bwd_fast_workspace_bytes(uint32_t C, uint32_t N, int64_t n_isects);

}  // namespace gsx
