// gsx_frontend.hip — the per-Gaussian front end of one `--gut` render in ONE streaming kernel (one camera):
//
//     SplatData activations (splat_data.cpp:267-286)  ->  projection_ut_3dgs_fused (ProjectionUT3DGSFused.cu:16-203)
//       ->  SH colours (+0.5, clamp_min 0: rasterizer.cpp:248-262, SphericalHarmonicsCUDA.cu:373-399)
//       ->  the packed 64 B camera-space record the blend kernels stage per tile intersection (gsx_record.hpp)
//
// The reference runs these as four groups of launches over the same Gaussians (activations, projection, SH + two elementwise ops, and the
// per-batch set-up inside the blend kernel); gsx used to mirror that with four per-Gaussian kernels (activations + projection, SH colours,
// pack_records, ~470 B of HBM traffic per Gaussian).  Here one lane owns one Gaussian from its raw parameters to its record:
//     reads   means 12 + scaling_raw 12 + rotation_raw 16 + opacity_raw 4 + the active SH bases 12 (deg+1)^2     (236 B at degree 3)
//     writes  scales 12 + quats 16 + opacities 4 (the activated copies the backward and the operators read) + radii 8 + means2d 8 +
//             depths 4 + conics 12 (optional: the render path passes NULL, nothing downstream reads them) + colours 12 + record 64
//             (140 B) + the four empty chain heads of the backward's record lists, 16 B
// = 376 B per visible Gaussian at degree 3 (+ 16 B of heads); a culled Gaussian stops after the projection (84 B: its SH row is never fetched, its record
// never written — no tile list can name it).  Every value is computed by the same device functions, in the same order of operations, as
// the separate operators (ut_project, ShBasis::eval, store_packed_record): activated parameters, projection and colours are bit-identical;
// the records agree to the last bit or two (the compiler contracts a few a*b+c of make_record differently in the two kernels) (tests/test_gpu_fused.py).
// HBM-bound streaming; the coefficient rows are fetched AFTER the projection (rows of culled Gaussians never), through LDS where the
// layout allows it (see frontend_kernel).
#include "gsx_record.hpp"
#include "gsx_sh_basis.hpp"
#include "gsx_ut_project.hpp"

namespace gsx {

void set_error(const char* msg);
int check_launch(const char* what);
const char* test_switch(const char* name);
size_t raster_fwd_fast_workspace_bytes(uint32_t C, uint32_t N);
int32_t* raster_fwd_fast_heads(const float4* packed, uint32_t C, uint32_t N);
uint32_t* raster_fwd_fast_alloc(const float4* packed, uint32_t C, uint32_t N);

constexpr int FE_BLOCK = 256;

struct FrontendOut {
    float* scales; float* quats; float* opacities;                   // activated copies [N,3] [N,4] [N]
    int32_t* radii; float* means2d; float* depths; float* conics;   // projection [1,N,2] [1,N,2] [1,N] [1,N,3]
    float* colors;                                                   // [1,N,3]
    float4* packed;                                                  // [N] x 64 B records
    int32_t* heads;                                                  // head planes of the backward's records (gsx_raster_common.hpp), RANGE mode: plane 0 [N] first slot, plane 1 [N] cursor
    uint32_t* alloc;                                                 // word [1] = mode of the planes; from word 64 on: the waves' slot totals (ranges), scanned into first slots behind this kernel
    int ranges;                                                      // 1: plane 0 = offset of the Gaussian's record run inside its wave, plane 1 = 0; 0: the four planes = -1 (empty chains)
};

// STAGE != 0: the wave's 64 coefficient rows — one contiguous 64 * K * 12 B span — are streamed into LDS with fully coalesced 16 B / lane
// loads and every lane then reads ITS row from LDS (pitch 52 floats: a quarter wave's ds_read_b128 cover the 64 banks exactly once),
// instead of twelve 192 B-strided 16 B loads per lane that pull every line through L1 twelve times.  STAGE == 2 passes the rows through
// a 32-row tile in two halves (6.6 KB of LDS per wave instead of 13.3: 4 instead of 3 waves per SIMD).  Measured in the training step
// (S-1M, bench per-op events, same box): direct loads 0.0915 ms, whole tile 0.084, two halves 0.079 (= 4.8 TB/s; four quarters 0.079).
// Taken when the active bases are the whole row and rows are multiples of 16 B (K = (deg + 1)^2, K * 3 % 4 == 0: degree 3 with K = 16 —
// two halves —, degree 1 with K = 4 — whole tile); otherwise the lane loads the active part of its row itself.  Rows of culled Gaussians
// are not fetched either way.  GSX_FE_STAGE=0|1|2 (test switch) forces a variant.
// (Round 4, not kept: the rows in two COLUMN halves — bases 0..7, then 8..15, the colour accumulated per half in the same order, bit-identical —
// leaves 24 instead of 48 coefficient registers live: 96 VGPRs, 5 instead of 4 waves per SIMD, and 0.081 -> 0.090 ms: the 96 B half-row runs
// use half of every 192 B the memory system moves per pass; occupancy is not what bounds this kernel.)
template <int KIND, int DEG, int STAGE>
__global__ __launch_bounds__(FE_BLOCK) void frontend_kernel(uint32_t N, uint32_t K, const float* __restrict__ means,
                                                            const float* __restrict__ rotation_raw, const float* __restrict__ scaling_raw,
                                                            const float* __restrict__ opacity_raw, const float* __restrict__ coeffs,
                                                            gsx_cameras cams, uint32_t W, uint32_t H, float eps2d, float near_plane,
                                                            float far_plane, float radius_clip, gsx_ut_params ut, FrontendOut out) {
    constexpr int NB = (DEG + 1) * (DEG + 1);
    constexpr int NB3 = NB * 3;
    constexpr int NQ = (NB3 + 3) / 4;
    constexpr int LQ = NQ + 1;   // LDS row pitch in float4 (STAGE)
    constexpr int TROWS = STAGE == 2 ? 32 : 64;   // rows of the wave's tile (STAGE == 2: the wave's rows pass through it in two halves)
    __shared__ float4 s_rows[STAGE ? (FE_BLOCK / 64) * TROWS * LQ : 1];
    const uint32_t gid = blockIdx.x * FE_BLOCK + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const bool in_range = gid < N;
    const Camera<KIND> cam(cams, 0, W, H);
    const ShutterPoses sp(cams.viewmats0, nullptr);

    f3 mean{0.f, 0.f, 0.f}, scale{1.f, 1.f, 1.f};
    float4 q_act = make_float4(1.f, 0.f, 0.f, 0.f);
    float opacity = 0.f;
    UtProjOut p;
    bool visible = false;
    if (in_range) {
        // ---- activations (projection_ut_kernel<KIND, true>, verbatim) ----
        mean = {means[(size_t)gid * 3], means[(size_t)gid * 3 + 1], means[(size_t)gid * 3 + 2]};
        scale = {scaling_raw[(size_t)gid * 3], scaling_raw[(size_t)gid * 3 + 1], scaling_raw[(size_t)gid * 3 + 2]};
        quat q{rotation_raw[(size_t)gid * 4], rotation_raw[(size_t)gid * 4 + 1], rotation_raw[(size_t)gid * 4 + 2], rotation_raw[(size_t)gid * 4 + 3]};
        opacity = opacity_raw[gid];
        scale = {expf(scale.x), expf(scale.y), expf(scale.z)};
        const float inv = 1.f / fmaxf(sqrtf(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z), 1e-12f);
        q = {q.w * inv, q.x * inv, q.y * inv, q.z * inv};
        opacity = 1.f / (1.f + expf(-opacity));
        out.scales[(size_t)gid * 3] = scale.x; out.scales[(size_t)gid * 3 + 1] = scale.y; out.scales[(size_t)gid * 3 + 2] = scale.z;
        q_act = make_float4(q.w, q.x, q.y, q.z);
        reinterpret_cast<float4*>(out.quats)[gid] = q_act;
        out.opacities[gid] = opacity;
        {   // glm::normalize(quat)
            const float len = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
            if (len <= 0.f) q = {1.f, 0.f, 0.f, 0.f};
            else { const float o = 1.f / len; q = {q.w * o, q.x * o, q.y * o, q.z * o}; }
        }
        // ---- projection ----
        visible = ut_project<KIND, true>(cam, sp, mean, scale, q, true, opacity, W, H, eps2d, near_plane, far_plane, radius_clip, ut, p);
    }
    // ---- the rectangle of 16-pixel tiles the intersection derives from (means2d, radii); on frames of large footprints (out.ranges) the
    //      backward's records of a Gaussian get that many consecutive slots: offset inside the wave here, the waves' totals scanned later ----
    uint32_t rect_x = 0u, rect_y = 0u;
    int32_t head0 = -1;   // chains: an empty chain
    if (visible) {
        rect_x = tile16_range(p.im.x, (float)(int32_t)p.radius_x, (W + 15u) / 16u);
        rect_y = tile16_range(p.im.y, (float)(int32_t)p.radius_y, (H + 15u) / 16u);
    }
    if (out.ranges) {   // (uniform; all 64 lanes are alive here: no lane has returned yet)
        const uint32_t n_slots = visible ? ((rect_x >> 16) - (rect_x & 0xFFFFu)) * ((rect_y >> 16) - (rect_y & 0xFFFFu)) : 0u;
        const uint32_t incl = wave_incl_scan_u32(n_slots);
        if (lane == 63u) out.alloc[64u + (blockIdx.x * FE_BLOCK + threadIdx.x) / 64u] = incl;
        head0 = (int32_t)(incl - n_slots);
    }
    if (gid == 0u) out.alloc[1] = out.ranges ? 1u : 0u;   // REC_MODE_RANGES / REC_MODE_CHAINS

    float row[NQ * 4];
    if (STAGE) {
        // ---- the wave's coefficient rows -> LDS -> this lane's registers (only the rows of visible Gaussians are fetched) ----
        const unsigned long long live = __builtin_amdgcn_ballot_w64(visible);
        if (live != 0ull) {
            const uint32_t e0 = blockIdx.x * FE_BLOCK + wave * 64u;
            const uint32_t rows = min(64u, N - e0);
            const float4* src = reinterpret_cast<const float4*>(coeffs + (size_t)e0 * K * 3u);
            float4* tile = s_rows + wave * (uint32_t)(TROWS * LQ);
            constexpr int NH = 64 / TROWS, NL = NQ / NH;   // passes, loads per lane and pass
            static_assert(NQ % NH == 0, "rows split into equal passes");
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                float4 v[NL];
#pragma unroll
                for (int i = 0; i < NL; ++i) {   // NL loads of 16 B per lane issued back to back
                    const uint32_t j = lane + 64u * (uint32_t)i;            // float4 index inside this pass's TROWS rows
                    const uint32_t e = j / (uint32_t)NQ + (uint32_t)(h * TROWS);
                    const bool ok = e < rows && ((live >> (e & 63u)) & 1ull) != 0ull;
                    v[i] = ok ? src[(size_t)h * TROWS * NQ + j] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int i = 0; i < NL; ++i) {   // (rows of culled Gaussians are written as zeros: nobody reads them)
                    const uint32_t j = lane + 64u * (uint32_t)i;
                    const uint32_t el = j / (uint32_t)NQ;
                    tile[el * (uint32_t)LQ + (j - el * (uint32_t)NQ)] = v[i];
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the tile is private to this wave: no block barrier
                __builtin_amdgcn_wave_barrier();
                if ((int)(lane / TROWS) == h) {
                    const float4* rq = tile + (lane % TROWS) * LQ;
#pragma unroll
                    for (int k = 0; k < NQ; ++k) {
                        const float4 r4 = rq[k];
                        row[k * 4] = r4.x; row[k * 4 + 1] = r4.y; row[k * 4 + 2] = r4.z; row[k * 4 + 3] = r4.w;
                    }
                }
                if (h + 1 < NH) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
    }
    if (!in_range) return;
    out.heads[gid] = head0;                        // chains: an empty chain; ranges: offset of the run inside the wave
    if (out.ranges) {
        out.heads[(size_t)N + gid] = 0;            // ranges: plane 1 = records claimed so far
    } else {
#pragma unroll
        for (int k = 1; k < 4; ++k) out.heads[(size_t)k * N + gid] = -1;   // chains: the backward may use all four planes (gsx_raster_common.hpp: NSUB)
    }
    float* col = out.colors + (size_t)gid * 3;
    if (!visible) {
        out.radii[(size_t)gid * 2] = 0; out.radii[(size_t)gid * 2 + 1] = 0;   // as upstream, only radii is written for a culled Gaussian
        col[0] = 0.f; col[1] = 0.f; col[2] = 0.f;                             // masked SH row (sh_colors_fwd writes zeros)
        store_null_record(out.packed + (size_t)gid * 4);                      // (no list names a culled Gaussian; the workspace still never holds garbage)
        return;
    }
    reinterpret_cast<int2*>(out.radii)[gid] = make_int2((int32_t)p.radius_x, (int32_t)p.radius_y);
    reinterpret_cast<float2*>(out.means2d)[gid] = make_float2(p.im.x, p.im.y);
    out.depths[gid] = p.depth;
    if (out.conics) {   // optional: nothing downstream of the render path reads them
        out.conics[(size_t)gid * 3] = p.c11 * p.ood;
        out.conics[(size_t)gid * 3 + 1] = -p.c01 * p.ood;
        out.conics[(size_t)gid * 3 + 2] = p.c00 * p.ood;
    }

    // ---- SH colours (sh_colors_fwd_direct_kernel, verbatim) ----
    if (!STAGE) {
        const float4* src = reinterpret_cast<const float4*>(coeffs + (size_t)gid * K * 3u);
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            const float4 v = src[k];
            row[k * 4] = v.x; row[k * 4 + 1] = v.y; row[k * 4 + 2] = v.z; row[k * 4 + 3] = v.w;
        }
    }
    const f3 cp = cam_position(cams.viewmats0);
    float x = mean.x - cp.x, y = mean.y - cp.y, z = mean.z - cp.z;
    if (DEG >= 1) {
        const float inorm = rsqrtf(x * x + y * y + z * z);
        x *= inorm; y *= inorm; z *= inorm;
    }
    float Y[NB];
    ShBasis<DEG>::template eval<false>(x, y, z, Y, nullptr, nullptr, nullptr);
    float cr = 0.f, cg = 0.f, cb = 0.f;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        cr += Y[k] * row[k * 3];
        cg += Y[k] * row[k * 3 + 1];
        cb += Y[k] * row[k * 3 + 2];
    }
    RawG raw;
    raw.rgb = {fmaxf(cr + 0.5f, 0.f), fmaxf(cg + 0.5f, 0.f), fmaxf(cb + 0.5f, 0.f)};
    col[0] = raw.rgb.x; col[1] = raw.rgb.y; col[2] = raw.rgb.z;

    // ---- packed record (pack_records_kernel, verbatim: from the ACTIVATED parameters, as the blend operators receive them) ----
    raw.g = (int32_t)gid;
    raw.mu = mean;
    raw.q = q_act;
    raw.sc = scale;
    raw.opac = opacity;
    const CamFrame cf = make_cam_frame(sp);
    // + the rectangle of 16-pixel tiles the intersection derives from (means2d, radii): read by the blend only under 32-pixel lists
    store_packed_record(raw, cf, out.packed + (size_t)gid * 4, rect_x, rect_y);
}

// the waves' slot totals -> first slot of every wave (exclusive scan in place, one block: 15 625 waves at 1 M Gaussians)
__global__ __launch_bounds__(1024) void wave_first_slot_kernel(uint32_t n_waves, uint32_t* __restrict__ w) {
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0u;
    __syncthreads();
    for (uint32_t base = 0; base < n_waves; base += 1024u) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n_waves ? w[i] : 0u;
        const uint32_t incl = wave_incl_scan_u32(v);
        if (lane == 63u) s_wave[wave] = incl;
        __syncthreads();
        uint32_t before = s_carry;
        for (uint32_t k = 0; k < wave; ++k) before += s_wave[k];
        if (i < n_waves) w[i] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023u) s_carry = before + incl;
        __syncthreads();
    }
}

}  // namespace gsx

using namespace gsx;

extern "C" int gsx_frontend_fused_supported(uint32_t K, uint32_t degrees_to_use, const gsx_cameras* cams, const float* coeffs) {
    if (!cams || cams->C != 1 || cams->camera_model != GSX_CAMERA_PINHOLE || cams->shutter != GSX_SHUTTER_GLOBAL || cams->viewmats1) return 0;
    if (degrees_to_use > 4 || (degrees_to_use + 1) * (degrees_to_use + 1) > K) return 0;
    const uint32_t nq4 = (((degrees_to_use + 1) * (degrees_to_use + 1) * 3 + 3) / 4) * 4;
    return ((K * 3u) % 4u == 0u && nq4 <= K * 3u && (((uintptr_t)coeffs) & 15u) == 0) ? 1 : 0;   // rows of whole, aligned 16 B vectors
}

extern "C" int gsx_frontend_fused(uint32_t N, uint32_t K, uint32_t degrees_to_use, const float* means, const float* rotation_raw,
                                  const float* scaling_raw, const float* opacity_raw, const float* coeffs, const gsx_cameras* cams,
                                  uint32_t image_width, uint32_t image_height, float eps2d, float near_plane, float far_plane, float radius_clip,
                                  const gsx_ut_params* ut, float* scales, float* quats, float* opacities, int32_t* radii, float* means2d,
                                  float* depths, float* conics, float* colors, void* fwd_workspace, size_t workspace_bytes, int record_ranges, void* stream) {
    if (!cams || !ut) { set_error("frontend_fused: cams/ut is null"); return GSX_ERR_INVALID_ARGUMENT; }
    if (!gsx_frontend_fused_supported(K, degrees_to_use, cams, coeffs)) {
        set_error("frontend_fused: one global-shutter pinhole camera, SH rows of whole 16 B vectors (gsx_frontend_fused_supported)");
        return GSX_ERR_UNSUPPORTED;
    }
    if (N == 0) return GSX_OK;
    if (!means || !rotation_raw || !scaling_raw || !opacity_raw || !coeffs || !cams->viewmats0 || !cams->Ks || !scales || !quats || !opacities ||
        !radii || !means2d || !depths || !colors || !fwd_workspace) {
        set_error("frontend_fused: null pointer");
        return GSX_ERR_INVALID_ARGUMENT;
    }
    if (workspace_bytes < raster_fwd_fast_workspace_bytes(1, N)) { set_error("frontend_fused: workspace too small (gsx_rasterize_fwd_workspace_bytes)"); return GSX_ERR_WORKSPACE_TOO_SMALL; }
    float4* packed_base = (float4*)(((uintptr_t)fwd_workspace + 255) & ~(uintptr_t)255);   // the blend forward's workspace layout
    FrontendOut out{scales, quats, opacities, radii, means2d, depths, conics, colors, packed_base, raster_fwd_fast_heads(packed_base, 1, N), raster_fwd_fast_alloc(packed_base, 1, N), record_ranges ? 1 : 0};
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((N + FE_BLOCK - 1) / FE_BLOCK), block(FE_BLOCK);
    const bool distorted = cams->radial || cams->tangential || cams->thin_prism;
    // rows staged through LDS when the active bases are the whole, 16 B-multiple row (see frontend_kernel); GSX_FE_STAGE=0 (test switch): never
    const uint32_t nb = (degrees_to_use + 1) * (degrees_to_use + 1);
    const char* sw = test_switch("GSX_FE_STAGE");
    const int stage = (nb == K && (K * 3u) % 4u == 0u) ? ((sw && sw[0] >= '0' && sw[0] <= '2') ? sw[0] - '0' : 2) : 0;
#define GSX_FE(KIND, D, S)                                                                                                                  \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(frontend_kernel<KIND, D, S>), grid, block, 0, st, N, K, means, rotation_raw, scaling_raw, opacity_raw, coeffs, \
                       *cams, image_width, image_height, eps2d, near_plane, far_plane, radius_clip, *ut, out)
#define GSX_FE_ST(KIND, D) do { if (stage == 2) GSX_FE(KIND, D, 2); else if (stage == 1) GSX_FE(KIND, D, 1); else GSX_FE(KIND, D, 0); } while (0)
#define GSX_FE_DEG(KIND)                                                                                       \
    switch (degrees_to_use) { case 0: GSX_FE(KIND, 0, 0); break;                                                    \
                              case 1: if (stage) GSX_FE(KIND, 1, 1); else GSX_FE(KIND, 1, 0); break;           \
                              case 2: GSX_FE(KIND, 2, 0); break; case 3: GSX_FE_ST(KIND, 3); break;            \
                              default: GSX_FE(KIND, 4, 0); break; }
    if (distorted) { GSX_FE_DEG(CAM_OPENCV_PINHOLE) } else { GSX_FE_DEG(CAM_PERFECT_PINHOLE) }
#undef GSX_FE_DEG
#undef GSX_FE_ST
#undef GSX_FE
    if (record_ranges) hipLaunchKernelGGL(wave_first_slot_kernel, dim3(1), dim3(1024), 0, st, (N + 63u) / 64u, out.alloc + 64);
    return check_launch("frontend_fused");
}
