// gsx_raster.hip — world-space (3DGUT) front-to-back alpha compositing for gfx950: forward and backward.
//
// Replaces gsplat::rasterize_to_pixels_from_world_3dgs_{fwd,bwd} (reference: gsplat/Rasterization.cpp:20-261,
// kernels gsplat/RasterizeToPixelsFromWorld3DGSFwd.cu:19-279 and ...Bwd.cu:16-373, helpers gsplat/Utils.cuh).
//
// Mapping: one 256-thread workgroup (4 wave64) per 16x16 tile; wave w owns the 8x8 quadrant
// (w&1, w>>1) of the tile, lane l the pixel (l&7, l>>3) inside it.  The tile's depth-sorted Gaussians
// are processed in chunks of 256: every thread gathers one Gaussian through flatten_ids, turns it
// into the 16-float record the pixel loop needs (M = diag(1/s) R^T, gro = M (o - mu), opacity, rgb)
// and parks it in LDS; the pixel loop then reads records at a wave-uniform index (LDS broadcast).
//
// Algebra kept identical to the reference, with one hoist that is exact for a global shutter: the ray
// origin o is the camera centre for every pixel (Cameras.cuh:261-265), so gro = M (o - mu) is a
// per-Gaussian quantity computed once at staging instead of once per (pixel, Gaussian).
//
// Backward: the reference reduces 14 gradient floats over each 32-lane warp and issues 14 global
// atomics per (warp, Gaussian) = 112 per (tile, Gaussian).  Here every lane produces the 16 sums that
// are linear in the per-pixel terms (v_rgb[3], v_opacity, v_gro[3], v_grd (x) d [9]); they are reduced
// over the wave with DPP row operations, added into a per-chunk LDS accumulator (ds_add_f32), and
// after the chunk ONE thread per Gaussian applies the mean / quaternion / scale chain rule
// (Utils.cuh:104-158) and issues the 14 global atomics: 14 per (tile, Gaussian), 8x fewer.
#include <stdlib.h>
#include <string.h>

#include "gsx_raster_common.hpp"

namespace gsx {

// One staged Gaussian: 4 x float4 = 64 B.
//  r0 = (M00 M01 M02 gro.x)  r1 = (M10 M11 M12 gro.y)  r2 = (M20 M21 M22 gro.z)  r3 = (opac, r, g, b)
// When the ray origin is per pixel (rolling shutter) the gro slots hold mu instead.
struct Staged { float4 r0, r1, r2, r3; };

template <bool HOIST>
GSX_DEV void stage_gaussian(const RasterArgs& a, int32_t g, f3 org, Staged& s) {
    const int32_t gi = (a.C == 1) ? g : (int32_t)((uint32_t)g % a.N);  // params are [N]; colours/opacities [C,N]
    const f3 mu{a.means[(size_t)gi * 3], a.means[(size_t)gi * 3 + 1], a.means[(size_t)gi * 3 + 2]};
    const float4 q = reinterpret_cast<const float4*>(a.quats)[gi];
    const f3 sc{a.scales[(size_t)gi * 3], a.scales[(size_t)gi * 3 + 1], a.scales[(size_t)gi * 3 + 2]};
    const m33 R = quat_to_rotmat(q.x, q.y, q.z, q.w);
    const float is0 = 1.f / sc.x, is1 = 1.f / sc.y, is2 = 1.f / sc.z;
    // M(r,c) = (1/s_r) R(c,r)
    s.r0 = make_float4(is0 * R.a[0][0], is0 * R.a[1][0], is0 * R.a[2][0], 0.f);
    s.r1 = make_float4(is1 * R.a[0][1], is1 * R.a[1][1], is1 * R.a[2][1], 0.f);
    s.r2 = make_float4(is2 * R.a[0][2], is2 * R.a[1][2], is2 * R.a[2][2], 0.f);
    if (HOIST) {
        const f3 d = org - mu;
        s.r0.w = s.r0.x * d.x + s.r0.y * d.y + s.r0.z * d.z;
        s.r1.w = s.r1.x * d.x + s.r1.y * d.y + s.r1.z * d.z;
        s.r2.w = s.r2.x * d.x + s.r2.y * d.y + s.r2.z * d.z;
    } else {
        s.r0.w = mu.x; s.r1.w = mu.y; s.r2.w = mu.z;
    }
    s.r3 = make_float4(a.opacities[g], a.colors[(size_t)g * 3], a.colors[(size_t)g * 3 + 1], a.colors[(size_t)g * 3 + 2]);
}

// alpha of one (pixel, Gaussian) pair, reference order (Fwd.cu:227-239)
template <bool HOIST>
GSX_DEV float pair_alpha(const Staged& s, f3 ray_o, f3 ray_d, f3& gro, f3& grd, f3& grd_n, f3& gc, float& vis, float& il) {
    if (HOIST) gro = {s.r0.w, s.r1.w, s.r2.w};
    else {
        const f3 d = ray_o - f3{s.r0.w, s.r1.w, s.r2.w};
        gro = {s.r0.x * d.x + s.r0.y * d.y + s.r0.z * d.z, s.r1.x * d.x + s.r1.y * d.y + s.r1.z * d.z,
               s.r2.x * d.x + s.r2.y * d.y + s.r2.z * d.z};
    }
    grd = {s.r0.x * ray_d.x + s.r0.y * ray_d.y + s.r0.z * ray_d.z, s.r1.x * ray_d.x + s.r1.y * ray_d.y + s.r1.z * ray_d.z,
           s.r2.x * ray_d.x + s.r2.y * ray_d.y + s.r2.z * ray_d.z};
    const float l = dot3(grd, grd);
    il = l > 0.f ? rsqrtf(l) : 1.f;
    grd_n = grd * il;
    gc = cross3(grd_n, gro);
    const float power = -0.5f * dot3(gc, gc);
    vis = __expf(power);
    return fminf(0.999f, s.r3.x * vis);
}

// ---- wave-level culling for the reference-order kernels (any camera model, any shutter) ----------------------------------------------
// alpha >= 1/255 needs |grd_n x gro|^2 <= 2 ln(255 o): the distance from the Gaussian's centre to the ray, measured in the whitened
// frame M = diag(1/s) R^T, which is at least (Euclidean distance) / s_max.  So a Gaussian whose bounding sphere of radius
// s_max sqrt(2 ln(255 o)) misses the DOUBLE CONE that bounds the 64 ray lines of a wave contributes to none of its pixels.  The cone (apex = mean
// ray origin, axis = mean direction, half angle = largest deviation, plus the spread of the origins for a rolling shutter) is built
// once per wave; one lane tests one staged Gaussian (~15 VALU), the wave walks the survivors.  Conservative by construction.
struct WaveCone { f3 apex, axis; float cos_g, sin_g, spread; bool any; };

GSX_DEV float wave_all_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
GSX_DEV float wave_all_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
GSX_DEV float wave_all_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

GSX_DEV WaveCone make_wave_cone(bool active, f3 ray_o, f3 ray_d) {
    WaveCone c;
    const float n = wave_all_sum(active ? 1.f : 0.f);
    c.any = n > 0.f;
    const float inv = c.any ? 1.f / n : 0.f;
    c.apex = {wave_all_sum(active ? ray_o.x : 0.f) * inv, wave_all_sum(active ? ray_o.y : 0.f) * inv, wave_all_sum(active ? ray_o.z : 0.f) * inv};
    f3 ax{wave_all_sum(active ? ray_d.x : 0.f), wave_all_sum(active ? ray_d.y : 0.f), wave_all_sum(active ? ray_d.z : 0.f)};
    const float l = sqrtf(dot3(ax, ax));
    if (!(l > 1e-6f)) { c.axis = {0.f, 0.f, 1.f}; c.cos_g = -1.f; c.sin_g = 0.f; c.spread = INFINITY; return c; }  // degenerate bundle: never cull
    c.axis = ax * (1.f / l);
    const float dl = sqrtf(dot3(ray_d, ray_d));
    const float cosd = active ? dot3(ray_d, c.axis) / fmaxf(dl, 1e-20f) : 1.f;
    c.cos_g = fmaxf(-1.f, wave_all_min(cosd) - 1e-6f);
    c.sin_g = sqrtf(fmaxf(0.f, 1.f - c.cos_g * c.cos_g));
    const f3 od = ray_o - c.apex;
    c.spread = wave_all_max(active ? sqrtf(dot3(od, od)) : 0.f);
    return c;
}

// true when the sphere (centre mu, radius rad) may touch the cone
GSX_DEV bool cone_hits_sphere(const WaveCone& c, f3 mu, float rad) {
    if (!(rad >= 0.f)) return false;              // opacity <= 1/255: never visible (rad = -1)
    const f3 v = mu - c.apex;
    const float along = dot3(v, c.axis);
    const float perp = sqrtf(fmaxf(0.f, dot3(v, v) - along * along));
    // distance from the point to the cone's surface along the surface normal (negative inside): perp cos g - |along| sin g.  The
    // reference's alpha measures the distance to the infinite ray LINE (|grd_n x gro|, Fwd.cu:232-236), so a Gaussian on the backward
    // extension of the rays contributes too (wide fisheye, spread origins): the test is against the DOUBLE cone, hence |along|.
    return perp * c.cos_g - fabsf(along) * c.sin_g <= (rad + c.spread) * 1.0001f + 1e-6f;
}

// (mu, bounding radius) of a staged Gaussian
GSX_DEV float4 cull_sphere(const RasterArgs& a, int32_t g) {
    const int32_t gi = (a.C == 1) ? g : (int32_t)((uint32_t)g % a.N);
    const float smax = fmaxf(fmaxf(a.scales[(size_t)gi * 3], a.scales[(size_t)gi * 3 + 1]), a.scales[(size_t)gi * 3 + 2]);
    const float o = a.opacities[g];
    const float t2 = 2.f * __logf(255.f * o);
    const float rad = (t2 > 0.f) ? smax * sqrtf(t2) * 1.001f : -1.f;
    return make_float4(a.means[(size_t)gi * 3], a.means[(size_t)gi * 3 + 1], a.means[(size_t)gi * 3 + 2], (rad == rad) ? rad : INFINITY);
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <int KIND, bool HOIST>
__global__ __launch_bounds__(RB) void raster_fwd_kernel(RasterArgs a, float* __restrict__ render_colors,
                                                        float* __restrict__ render_alphas,
                                                        int32_t* __restrict__ last_ids, const uint8_t* __restrict__ only_tiles) {
    __shared__ Staged s_g[RB];
    __shared__ float4 s_sph[RB];
    const uint32_t tile_x = blockIdx.x, tile_y = blockIdx.y, cid = blockIdx.z;
    const uint32_t tile_id = tile_y * a.tw + tile_x;
    if (only_tiles != nullptr && !only_tiles[(size_t)cid * a.th * a.tw + tile_id]) return;  // the fast path rendered this tile
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    uint32_t i, j;
    thread_pixel(tid, tile_x, tile_y, i, j);
    const bool inside = i < a.H && j < a.W;
    const size_t pix = (size_t)cid * a.H * a.W + (size_t)i * a.W + j;
    const float* bg = a.backgrounds ? a.backgrounds + cid * 3 : nullptr;

    if (a.masks != nullptr && !a.masks[(size_t)cid * a.th * a.tw + tile_id]) {  // Fwd.cu:143-150
        if (inside)
            for (int k = 0; k < 3; ++k) render_colors[pix * 3 + k] = bg ? bg[k] : 0.f;
        return;
    }

    const Camera<KIND> cam(a.cams, cid, a.W, a.H);
    const ShutterPoses sp(a.cams.viewmats0 + cid * 16, a.cams.viewmats1 ? a.cams.viewmats1 + cid * 16 : nullptr);
    f3 ray_o, ray_d;
    const bool ray_ok = cam.pixel_to_world_ray(f2{(float)j + 0.5f, (float)i + 0.5f}, sp, ray_o, ray_d);
    f3 cam_org = ray_o;
    if (HOIST) {  // global shutter: identical origin for every pixel; recompute it where the ray was invalid
        const m33 Rinv = quat_to_mat_raw(quat_conj_over_norm2(sp.q0));
        const f3 rt = mul(Rinv, sp.t0);
        cam_org = {-rt.x, -rt.y, -rt.z};
    }
    bool done = !inside || !ray_ok;
    const WaveCone cone = make_wave_cone(!done, ray_o, ray_d);

    const int32_t* toff = a.tile_offsets + (size_t)cid * a.th * a.tw;
    const int32_t range_start = toff[tile_id];
    bool lists_ok;
    const int32_t lists_end = lists_total(a, lists_ok);
    const int32_t range_end = !lists_ok ? range_start : ((cid == a.C - 1 && tile_id == a.tw * a.th - 1) ? lists_end : toff[tile_id + 1]);
    const int32_t n_chunks = (range_end - range_start + RB - 1) / RB;

    float T = 1.f;
    uint32_t cur_idx = 0;
    float out_r = 0.f, out_g = 0.f, out_b = 0.f;
    for (int32_t b = 0; b < n_chunks; ++b) {
        if (__syncthreads_and(done)) break;  // Fwd.cu:188-190
        const int32_t chunk_start = range_start + RB * b;
        const int32_t idx = chunk_start + (int32_t)tid;
        if (idx < range_end) {
            const int32_t g = a.flatten_ids[idx];
            Staged s;
            stage_gaussian<HOIST>(a, g, cam_org, s);
            s_g[tid] = s;
            s_sph[tid] = cull_sphere(a, g);
        }
        __syncthreads();
        const int32_t chunk_size = min(RB, range_end - chunk_start);
        for (int32_t sub = 0; sub < chunk_size; sub += 64) {
            // one candidate per lane against the cone of this wave's rays, then the wave walks the survivors front to back
            bool hit = false;
            if (cone.any && sub + (int32_t)lane < chunk_size) {
                const float4 sp4 = s_sph[sub + lane];
                hit = cone_hits_sphere(cone, f3{sp4.x, sp4.y, sp4.z}, sp4.w);
            }
            unsigned long long todo = __builtin_amdgcn_ballot_w64(hit);
            while (todo) {
                const int32_t t = sub + __builtin_ctzll(todo);
                todo &= todo - 1ull;
                if (done) continue;
                const Staged s = s_g[t];
                f3 gro, grd, grd_n, gc; float vis, il;
                const float alpha = pair_alpha<HOIST>(s, ray_o, ray_d, gro, grd, grd_n, gc, vis, il);
                if (alpha < ALPHA_MIN) continue;
                const float next_T = T * (1.f - alpha);
                if (next_T <= 1e-4f) { done = true; continue; }
                const float w = alpha * T;
                out_r += s.r3.y * w; out_g += s.r3.z * w; out_b += s.r3.w * w;
                cur_idx = (uint32_t)(chunk_start + t);
                T = next_T;
            }
        }
    }
    if (inside) {
        render_alphas[pix] = 1.f - T;
        render_colors[pix * 3] = bg ? out_r + T * bg[0] : out_r;
        render_colors[pix * 3 + 1] = bg ? out_g + T * bg[1] : out_g;
        render_colors[pix * 3 + 2] = bg ? out_b + T * bg[2] : out_b;
        last_ids[pix] = (int32_t)cur_idx;
    }
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
constexpr int NACC = 16;  // v_rgb[3], v_opacity, v_gro[3], G = sum v_grd (x) d [9]

template <int KIND, bool HOIST>
__global__ __launch_bounds__(RB) void raster_bwd_kernel(RasterArgs a, const float* __restrict__ render_alphas,
                                                        const int32_t* __restrict__ last_ids,
                                                        const float* __restrict__ v_render_colors,
                                                        const float* __restrict__ v_render_alphas,
                                                        float* __restrict__ v_means, float* __restrict__ v_quats,
                                                        float* __restrict__ v_scales, float* __restrict__ v_colors,
                                                        float* __restrict__ v_opacities, const uint8_t* __restrict__ only_tiles,
                                                        float4* __restrict__ grad_rec, int32_t* __restrict__ grad_head) {
    __shared__ Staged s_g[RB];
    __shared__ float s_acc[RB * NACC];
    __shared__ int32_t s_id[RB];
    __shared__ int32_t s_touched[RB];
    __shared__ float4 s_sph[RB];
    __shared__ float4 s_omu[HOIST ? RB : 1];   // global shutter: camera centre - mean of the staged Gaussians (v_gro (x) (o - mu) is folded per PIXEL, below)
    const uint32_t tile_x = blockIdx.x, tile_y = blockIdx.y, cid = blockIdx.z;
    const uint32_t tile_id = tile_y * a.tw + tile_x;
    if (only_tiles != nullptr && !only_tiles[(size_t)cid * a.th * a.tw + tile_id]) return;  // the fast path handled this tile
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    if (a.masks != nullptr && !a.masks[(size_t)cid * a.th * a.tw + tile_id]) return;  // Bwd.cu:84-86
    uint32_t i, j;
    thread_pixel(tid, tile_x, tile_y, i, j);
    const bool inside = i < a.H && j < a.W;
    const size_t pix = (size_t)cid * a.H * a.W + (size_t)min(i, a.H - 1) * a.W + min(j, a.W - 1);
    const float* bg = a.backgrounds ? a.backgrounds + cid * 3 : nullptr;

    const Camera<KIND> cam(a.cams, cid, a.W, a.H);
    const ShutterPoses sp(a.cams.viewmats0 + cid * 16, a.cams.viewmats1 ? a.cams.viewmats1 + cid * 16 : nullptr);
    f3 ray_o, ray_d;
    const bool ray_ok = cam.pixel_to_world_ray(f2{(float)j + 0.5f, (float)i + 0.5f}, sp, ray_o, ray_d);
    f3 cam_org = ray_o;
    if (HOIST) {
        const m33 Rinv = quat_to_mat_raw(quat_conj_over_norm2(sp.q0));
        const f3 rt = mul(Rinv, sp.t0);
        cam_org = {-rt.x, -rt.y, -rt.z};
    }
    const bool active = inside && ray_ok;
    const WaveCone cone = make_wave_cone(active, ray_o, ray_d);

    const int32_t* toff = a.tile_offsets + (size_t)cid * a.th * a.tw;
    const int32_t range_start = toff[tile_id];
    bool lists_ok;
    const int32_t lists_end = lists_total(a, lists_ok);
    const int32_t range_end = !lists_ok ? range_start : ((cid == a.C - 1 && tile_id == a.tw * a.th - 1) ? lists_end : toff[tile_id + 1]);

    const float T_final = 1.f - render_alphas[pix];
    float T = T_final;
    float buf_r = 0.f, buf_g = 0.f, buf_b = 0.f;
    const int32_t bin_final = active ? last_ids[pix] : -1;
    const float vr = v_render_colors[pix * 3], vg = v_render_colors[pix * 3 + 1], vb = v_render_colors[pix * 3 + 2];
    const float va = v_render_alphas ? v_render_alphas[pix] : 0.f;
    float bg_dot = 0.f;
    if (bg) bg_dot = bg[0] * vr + bg[1] * vg + bg[2] * vb;

    // the last sorted index any pixel of the block needs: chunks entirely behind it are skipped
    __shared__ int32_t s_blockmax;
    if (tid == 0) s_blockmax = -1;
    __syncthreads();
    {
        int32_t m = bin_final;
        for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
        if (lane == 0) atomicMax(&s_blockmax, m);
    }
    __syncthreads();
    const int32_t block_last = min(s_blockmax, range_end - 1);
    if (block_last < range_start) return;
    const int32_t n_chunks = (block_last - range_start + RB) / RB;  // chunks counted from the back

    for (int32_t b = 0; b < n_chunks; ++b) {
        __syncthreads();  // previous chunk fully consumed
        const int32_t chunk_end = block_last - RB * b;               // inclusive, furthest-back first
        const int32_t chunk_size = min(RB, chunk_end + 1 - range_start);
        const int32_t idx = chunk_end - (int32_t)tid;
        if (idx >= range_start) {
            const int32_t g = a.flatten_ids[idx];
            Staged s;
            stage_gaussian<HOIST>(a, g, cam_org, s);
            s_g[tid] = s;
            s_id[tid] = g;
            s_sph[tid] = cull_sphere(a, g);
            if (HOIST) {
                const int32_t gi = (a.C == 1) ? g : (int32_t)((uint32_t)g % a.N);
                s_omu[tid] = make_float4(cam_org.x - a.means[(size_t)gi * 3], cam_org.y - a.means[(size_t)gi * 3 + 1], cam_org.z - a.means[(size_t)gi * 3 + 2], 0.f);
            }
        }
        s_touched[tid] = 0;
#pragma unroll
        for (int k = 0; k < NACC; ++k) s_acc[k * RB + tid] = 0.f;
        __syncthreads();

        for (int32_t sub = 0; sub < chunk_size; sub += 64) {
          bool hit = false;   // (same wave-cone test as the forward)
          if (cone.any && sub + (int32_t)lane < chunk_size) {
              const float4 sp4 = s_sph[sub + lane];
              hit = cone_hits_sphere(cone, f3{sp4.x, sp4.y, sp4.z}, sp4.w);
          }
          unsigned long long todo = __builtin_amdgcn_ballot_w64(hit);
          while (todo) {
            const int32_t t = sub + __builtin_ctzll(todo);
            todo &= todo - 1ull;
            const int32_t gidx = chunk_end - t;
            bool valid = active && gidx <= bin_final;
            f3 gro, grd, grd_n, gc; float vis = 0.f, il = 1.f, alpha = 0.f;
            Staged s;
            if (valid) {
                s = s_g[t];
                alpha = pair_alpha<HOIST>(s, ray_o, ray_d, gro, grd, grd_n, gc, vis, il);
                if (alpha < ALPHA_MIN) valid = false;
            }
            if (__builtin_amdgcn_ballot_w64(valid) == 0ull) continue;
            float acc[NACC];
#pragma unroll
            for (int k = 0; k < NACC; ++k) acc[k] = 0.f;
            if (valid) {
                const float ra = 1.f / (1.f - alpha);
                T *= ra;
                const float fac = alpha * T;
                acc[0] = fac * vr; acc[1] = fac * vg; acc[2] = fac * vb;
                float v_alpha = (s.r3.y * T - buf_r * ra) * vr + (s.r3.z * T - buf_g * ra) * vg + (s.r3.w * T - buf_b * ra) * vb;
                v_alpha += T_final * ra * va;
                if (bg) v_alpha += -T_final * ra * bg_dot;
                if (s.r3.x * vis <= 0.999f) {
                    const float v_vis = s.r3.x * v_alpha;
                    const float v_gd = -0.5f * vis * v_vis;
                    const f3 v_gc = gc * (2.f * v_gd);
                    const f3 cx = cross3(v_gc, gro);
                    const f3 v_grd_n{-cx.x, -cx.y, -cx.z};
                    const f3 v_gro = cross3(v_gc, grd_n);
                    // safe_normalize_bw (Utils.cuh:186-194)
                    const float il3 = il * il * il;
                    const f3 v_grd = v_grd_n * il - grd * (il3 * dot3(v_grd_n, grd));
                    acc[3] = vis * v_alpha;
                    acc[4] = v_gro.x; acc[5] = v_gro.y; acc[6] = v_gro.z;
                    acc[7] = v_grd.x * ray_d.x; acc[8] = v_grd.x * ray_d.y; acc[9] = v_grd.x * ray_d.z;
                    acc[10] = v_grd.y * ray_d.x; acc[11] = v_grd.y * ray_d.y; acc[12] = v_grd.y * ray_d.z;
                    acc[13] = v_grd.z * ray_d.x; acc[14] = v_grd.z * ray_d.y; acc[15] = v_grd.z * ray_d.z;
                    {
                        // v_Mt = v_grd (x) d + v_gro (x) (o - mu) per PIXEL, as Bwd.cu:325-326 has it (v_gro is kept for v_mean).  Rounds 1 - 4 folded
                        // the second product once per (tile, Gaussian) behind the sums when the origin is the same for every pixel: the two sums are
                        // |gro| ~ depth / scale times larger than their total, and on needle-shaped Gaussians (scale ratios of several hundred) the
                        // short axes' scale gradients lost 3e-3 to it (tests/test_gpu_reference_hip.py: regime "needles").  Pixel by pixel the two
                        // products cancel before they are summed.
                        f3 omu;
                        if (HOIST) { const float4 om = s_omu[t]; omu = {om.x, om.y, om.z}; }
                        else omu = ray_o - f3{s.r0.w, s.r1.w, s.r2.w};
                        acc[7] += v_gro.x * omu.x; acc[8] += v_gro.x * omu.y; acc[9] += v_gro.x * omu.z;
                        acc[10] += v_gro.y * omu.x; acc[11] += v_gro.y * omu.y; acc[12] += v_gro.y * omu.z;
                        acc[13] += v_gro.z * omu.x; acc[14] += v_gro.z * omu.y; acc[15] += v_gro.z * omu.z;
                    }
                }
                buf_r += s.r3.y * fac; buf_g += s.r3.z * fac; buf_b += s.r3.w * fac;
            }
            // 16 sums over the wave in one halving butterfly (~35 VALU instead of 16 x 6 DPP adds): 16 lanes end with one total each
            const float total = butterfly_reduce16(acc);
            if ((lane & 3u) == 0u) atomicAdd(&s_acc[butterfly_value_of_lane(lane) * RB + t], total);
            if (lane == 0u) s_touched[t] = 1;
          }
        }
        __syncthreads();

        // one thread per Gaussian of the chunk: chain rule to (mean, quat, scale) and 14 global atomics
        if ((int32_t)tid < chunk_size && s_touched[tid]) {
            const int32_t g = s_id[tid];
            const int32_t gi = (a.C == 1) ? g : (int32_t)((uint32_t)g % a.N);
            float A[NACC];
#pragma unroll
            for (int k = 0; k < NACC; ++k) A[k] = s_acc[k * RB + tid];
            float out[14];   // v_mean 3 | v_quat 4 | v_scale 3 | v_color 3 | v_opacity
            out[10] = A[0]; out[11] = A[1]; out[12] = A[2]; out[13] = A[3];
            const Staged s = s_g[tid];
            const f3 v_gro{A[4], A[5], A[6]};
            // v_mean = - M^T v_gro
            out[0] = -(s.r0.x * v_gro.x + s.r1.x * v_gro.y + s.r2.x * v_gro.z);
            out[1] = -(s.r0.y * v_gro.x + s.r1.y * v_gro.y + s.r2.y * v_gro.z);
            out[2] = -(s.r0.z * v_gro.x + s.r1.z * v_gro.y + s.r2.z * v_gro.z);
            // v_Mt(r,c) = sum over the pixels of v_grd_r d_c + v_gro_r (o - mu)_c   (Bwd.cu:325-326)
            float vMt[3][3] = {{A[7], A[8], A[9]}, {A[10], A[11], A[12]}, {A[13], A[14], A[15]}};
            const float4 qraw = reinterpret_cast<const float4*>(a.quats)[gi];
            const f3 sc{a.scales[(size_t)gi * 3], a.scales[(size_t)gi * 3 + 1], a.scales[(size_t)gi * 3 + 2]};
            // quat_scale_to_preci_half_vjp (Utils.cuh:128-158): v_M = v_Mt^T is dL/d(R S), S = diag(1/s)
            const float isv[3] = {1.f / sc.x, 1.f / sc.y, 1.f / sc.z};
            float w = qraw.x, x = qraw.y, y = qraw.z, z = qraw.w;
            const float inv_norm = rsqrtf(x * x + y * y + z * z + w * w);
            w *= inv_norm; x *= inv_norm; y *= inv_norm; z *= inv_norm;
            const m33 R = quat_to_mat_raw(quat{w, x, y, z});
            // v_R(r,c) = v_M(r,c) * is_c = vMt[c][r] * is_c ;  G(i,j) := glm v_R[i][j] = v_R(j,i) = vMt[i][j] * is_i
#define GSX_G(i, j) (vMt[i][j] * isv[i])
            float vq[4];
            vq[0] = 2.f * (x * (GSX_G(1, 2) - GSX_G(2, 1)) + y * (GSX_G(2, 0) - GSX_G(0, 2)) + z * (GSX_G(0, 1) - GSX_G(1, 0)));
            vq[1] = 2.f * (-2.f * x * (GSX_G(1, 1) + GSX_G(2, 2)) + y * (GSX_G(0, 1) + GSX_G(1, 0)) + z * (GSX_G(0, 2) + GSX_G(2, 0)) + w * (GSX_G(1, 2) - GSX_G(2, 1)));
            vq[2] = 2.f * (x * (GSX_G(0, 1) + GSX_G(1, 0)) - 2.f * y * (GSX_G(0, 0) + GSX_G(2, 2)) + z * (GSX_G(1, 2) + GSX_G(2, 1)) + w * (GSX_G(2, 0) - GSX_G(0, 2)));
            vq[3] = 2.f * (x * (GSX_G(0, 2) + GSX_G(2, 0)) + y * (GSX_G(1, 2) + GSX_G(2, 1)) - 2.f * z * (GSX_G(0, 0) + GSX_G(1, 1)) + w * (GSX_G(0, 1) - GSX_G(1, 0)));
#undef GSX_G
            const float qn[4] = {w, x, y, z};
            const float dq = vq[0] * qn[0] + vq[1] * qn[1] + vq[2] * qn[2] + vq[3] * qn[3];
#pragma unroll
            for (int k = 0; k < 4; ++k) out[3 + k] = (vq[k] - dq * qn[k]) * inv_norm;
            // v_scale[k] = -(1/s_k)^2 * sum_r R(r,k) * v_M(r,k) = -(is_k)^2 * sum_r R(r,k) * vMt[k][r]
#pragma unroll
            for (int k = 0; k < 3; ++k)
                out[7 + k] = -isv[k] * isv[k] * (R.a[0][k] * vMt[k][0] + R.a[1][k] * vMt[k][1] + R.a[2][k] * vMt[k][2]);
            if (grad_rec != nullptr) {
                // No float atomics reach memory: the 14 gradients of this (tile, Gaussian) go into ONE 64 B record at the intersection's
                // sorted index, chained per (camera, Gaussian) with a returning exchange; gsx_bwd_gather_grads_kernel sums the lists.
                // (On MI355X device-scope float atomics resolve memory-side: 14 of them per (tile, Gaussian) dominate this kernel.)
                const int32_t isect = chunk_end - (int32_t)tid;
                const int32_t prev = atomicExch(&grad_head[g], isect);
                float4* rec = grad_rec + (size_t)isect * 4;
                rec[0] = make_float4(out[0], out[1], out[2], out[3]);
                rec[1] = make_float4(out[4], out[5], out[6], out[7]);
                rec[2] = make_float4(out[8], out[9], out[10], out[11]);
                rec[3] = make_float4(out[12], out[13], 0.f, __int_as_float(prev));
            } else {
#pragma unroll
                for (int k = 0; k < 3; ++k) atomicAdd(&v_means[(size_t)gi * 3 + k], out[k]);
#pragma unroll
                for (int k = 0; k < 4; ++k) atomicAdd(&v_quats[(size_t)gi * 4 + k], out[3 + k]);
#pragma unroll
                for (int k = 0; k < 3; ++k) atomicAdd(&v_scales[(size_t)gi * 3 + k], out[7 + k]);
#pragma unroll
                for (int k = 0; k < 3; ++k) atomicAdd(&v_colors[(size_t)g * 3 + k], out[10 + k]);
                atomicAdd(&v_opacities[g], out[13]);
            }
        }
    }
}

// Sums the gradient records of every (camera, Gaussian) list built by raster_bwd_kernel in record mode and writes EVERY output
// element (no pre-zeroing): v_colors / v_opacities per camera, v_means / v_quats / v_scales summed over the cameras.
__global__ __launch_bounds__(256) void gsx_bwd_gather_grads_kernel(uint32_t C, uint32_t N, const float4* __restrict__ rec,
                                                                   const int32_t* __restrict__ head, float* __restrict__ v_means,
                                                                   float* __restrict__ v_quats, float* __restrict__ v_scales,
                                                                   float* __restrict__ v_colors, float* __restrict__ v_opacities) {
    const uint32_t gi = blockIdx.x * 256u + threadIdx.x;
    if (gi >= N) return;
    float geo[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) geo[k] = 0.f;
    for (uint32_t c = 0; c < C; ++c) {
        const size_t g = (size_t)c * N + gi;
        float col[4] = {0.f, 0.f, 0.f, 0.f};
        for (int32_t it = head[g]; it >= 0;) {
            const float4* r = rec + (size_t)it * 4;
            const float4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
            geo[0] += r0.x; geo[1] += r0.y; geo[2] += r0.z; geo[3] += r0.w;
            geo[4] += r1.x; geo[5] += r1.y; geo[6] += r1.z; geo[7] += r1.w;
            geo[8] += r2.x; geo[9] += r2.y; col[0] += r2.z; col[1] += r2.w;
            col[2] += r3.x; col[3] += r3.y;
            it = __float_as_int(r3.w);
        }
        v_colors[g * 3] = col[0]; v_colors[g * 3 + 1] = col[1]; v_colors[g * 3 + 2] = col[2];
        v_opacities[g] = col[3];
    }
    v_means[(size_t)gi * 3] = geo[0]; v_means[(size_t)gi * 3 + 1] = geo[1]; v_means[(size_t)gi * 3 + 2] = geo[2];
    reinterpret_cast<float4*>(v_quats)[gi] = make_float4(geo[3], geo[4], geo[5], geo[6]);
    v_scales[(size_t)gi * 3] = geo[7]; v_scales[(size_t)gi * 3 + 1] = geo[8]; v_scales[(size_t)gi * 3 + 2] = geo[9];
}

static int fill_args(RasterArgs& a, uint32_t N, int64_t n_isects, const float* means, const float* quats,
                     const float* scales, const float* colors, uint32_t channels, const float* opacities,
                     const float* backgrounds, const uint8_t* masks, uint32_t W, uint32_t H, uint32_t tile_size,
                     const gsx_cameras* cams, const int32_t* tile_offsets, const int32_t* flatten_ids, const char* who) {
    if (!cams) { set_error("rasterize_to_pixels_from_world_3dgs: cams is null"); return GSX_ERR_INVALID_ARGUMENT; }
    if (channels != 3) {  // Rasterization.cpp:65 asserts channels == 3
        set_error("rasterize_to_pixels_from_world_3dgs: Unsupported number of channels (only 3)");
        return GSX_ERR_INVALID_ARGUMENT;
    }
    // tile_size = the tile size the LISTS were built for: 16 (the reference's) or 32 (extension: one list per 2 x 2 pixel tiles, fast path only)
    if (tile_size != TILE && tile_size != 2 * TILE) { set_error("rasterize_to_pixels_from_world_3dgs: tile_size must be 16 (or 32: coarse lists, fast path)"); return GSX_ERR_UNSUPPORTED; }
    if (!means || !quats || !scales || !colors || !opacities || !tile_offsets || !cams->viewmats0 || !cams->Ks ||
        (n_isects > 0 && !flatten_ids)) {
        set_error("rasterize_to_pixels_from_world_3dgs: null pointer");
        return GSX_ERR_INVALID_ARGUMENT;
    }
    if (cams->camera_model != GSX_CAMERA_PINHOLE && cams->camera_model != GSX_CAMERA_FISHEYE) {
        set_error("rasterize_to_pixels_from_world_3dgs: unsupported camera model (ORTHO is rejected upstream too)");
        return GSX_ERR_UNSUPPORTED;
    }
    (void)who;
    a.C = cams->C; a.N = N; a.n_isects = n_isects;
    a.means = means; a.quats = quats; a.scales = scales; a.colors = colors; a.opacities = opacities;
    a.backgrounds = backgrounds; a.masks = masks;
    a.W = W; a.H = H; a.tw = (W + TILE - 1) / TILE; a.th = (H + TILE - 1) / TILE;
    a.lshift = tile_size == TILE ? 0u : 1u;
    a.ltw = (W + tile_size - 1) / tile_size; a.lth = (H + tile_size - 1) / tile_size;
    const char* rf = test_switch("GSX_LIST_RECT");   // tests only: "0" shows what the 32-px lists would composite without the rectangle filter
    a.rect_filter = (rf != nullptr && strcmp(rf, "0") == 0) ? 0u : 1u;
    a.chain_mask = 0u;
    a.rec_capacity = 0;
    a.cams = *cams;
    a.tile_offsets = tile_offsets; a.flatten_ids = flatten_ids;
    a.packed = nullptr;
    a.tile_flags = nullptr;
    a.lists_status = nullptr;
    a.n_isects_expected = n_isects;
    return GSX_OK;
}

// GSX_RASTER_PATH=generic forces the reference-order kernels (used by the tests to cover both paths)
static bool force_generic() {
    const char* e = test_switch("GSX_RASTER_PATH");
    return e != nullptr && strcmp(e, "generic") == 0;
}

static int cam_kind(const gsx_cameras& c) {
    if (c.camera_model == GSX_CAMERA_FISHEYE) return CAM_OPENCV_FISHEYE;
    return (c.radial || c.tangential || c.thin_prism) ? CAM_OPENCV_PINHOLE : CAM_PERFECT_PINHOLE;
}

}  // namespace gsx

using namespace gsx;

extern "C" int gsx_rasterize_to_pixels_from_world_3dgs_fwd(
    uint32_t N, int64_t n_isects, const float* means, const float* quats, const float* scales, const float* colors,
    uint32_t channels, const float* opacities, const float* backgrounds, const uint8_t* masks, uint32_t image_width,
    uint32_t image_height, uint32_t tile_size, const gsx_cameras* cams, const gsx_ut_params* ut,
    const int32_t* tile_offsets, const int32_t* flatten_ids, float* renders, float* alphas, int32_t* last_ids,
    void* workspace, size_t workspace_bytes, void* stream) {
    return gsx_rasterize_to_pixels_from_world_3dgs_fwd_packed(N, n_isects, means, quats, scales, colors, channels, opacities, backgrounds, masks,
                                                              image_width, image_height, tile_size, cams, ut, tile_offsets, flatten_ids, renders,
                                                              alphas, last_ids, workspace, workspace_bytes, 0, stream);
}

extern "C" int gsx_rasterize_to_pixels_from_world_3dgs_fwd_packed(
    uint32_t N, int64_t n_isects, const float* means, const float* quats, const float* scales, const float* colors,
    uint32_t channels, const float* opacities, const float* backgrounds, const uint8_t* masks, uint32_t image_width,
    uint32_t image_height, uint32_t tile_size, const gsx_cameras* cams, const gsx_ut_params* ut,
    const int32_t* tile_offsets, const int32_t* flatten_ids, float* renders, float* alphas, int32_t* last_ids,
    void* workspace, size_t workspace_bytes, int records_ready, void* stream) {
    return gsx_rasterize_to_pixels_from_world_3dgs_fwd_guarded(N, n_isects, means, quats, scales, colors, channels, opacities, backgrounds, masks,
                                                               image_width, image_height, tile_size, cams, ut, tile_offsets, flatten_ids, renders,
                                                               alphas, last_ids, workspace, workspace_bytes, records_ready, nullptr, n_isects, stream);
}

extern "C" int gsx_rasterize_to_pixels_from_world_3dgs_fwd_guarded(
    uint32_t N, int64_t n_isects, const float* means, const float* quats, const float* scales, const float* colors,
    uint32_t channels, const float* opacities, const float* backgrounds, const uint8_t* masks, uint32_t image_width,
    uint32_t image_height, uint32_t tile_size, const gsx_cameras* cams, const gsx_ut_params* ut,
    const int32_t* tile_offsets, const int32_t* flatten_ids, float* renders, float* alphas, int32_t* last_ids,
    void* workspace, size_t workspace_bytes, int records_ready, const int32_t* lists_status, int64_t n_isects_expected, void* stream) {
    (void)ut;
    RasterArgs a;
    int rc = fill_args(a, N, n_isects, means, quats, scales, colors, channels, opacities, backgrounds, masks, image_width,
                       image_height, tile_size, cams, tile_offsets, flatten_ids, "fwd");
    if (rc != GSX_OK) return rc;
    a.lists_status = lists_status;
    if (lists_status != nullptr && n_isects_expected > 0) a.n_isects_expected = n_isects_expected;
    if (!renders || !alphas || !last_ids) { set_error("rasterize fwd: null output"); return GSX_ERR_INVALID_ARGUMENT; }
    if (a.C == 0 || image_width == 0 || image_height == 0) return GSX_OK;
    const dim3 grid(a.tw, a.th, a.C), block(RB);
    hipStream_t st = (hipStream_t)stream;
    const bool hoist = cams->shutter == GSX_SHUTTER_GLOBAL;
    const int kind = cam_kind(*cams);
    const uint8_t* only_tiles = nullptr;
    // fast path: global shutter; a fisheye additionally needs its (camera, tile) flag plane to fit
    if (hoist && !force_generic() && workspace != nullptr && workspace_bytes >= raster_fwd_fast_workspace_bytes(a.C, a.N) &&
        (kind != CAM_OPENCV_FISHEYE || ((size_t)a.C * a.tw * a.th <= FAST_FLAG_BYTES && a.lshift == 0))) {
        // records_ready: gsx_frontend_fused already wrote the packed records of exactly these inputs into this workspace (pinholes only)
        only_tiles = launch_raster_fwd_fast(kind, a, renders, alphas, last_ids, workspace, workspace_bytes, st,
                                            records_ready != 0 && kind != CAM_OPENCV_FISHEYE);
        if (only_tiles == nullptr) return check_launch("rasterize_to_pixels_from_world_3dgs_fwd(fast)");
        // fisheye: tiles whose list holds a Gaussian without a usable chart were left to the reference-order kernel below
    }
    if (a.lshift) { set_error("rasterize fwd: lists per 32 x 32 pixels need the fast path (global-shutter pinhole, workspace, no GSX_RASTER_PATH=generic)"); return GSX_ERR_UNSUPPORTED; }
#define GSX_FWD(KIND)                                                                                                  \
    do {                                                                                                               \
        if (hoist) hipLaunchKernelGGL(HIP_KERNEL_NAME(raster_fwd_kernel<KIND, true>), grid, block, 0, st, a, renders, alphas, last_ids, only_tiles); \
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(raster_fwd_kernel<KIND, false>), grid, block, 0, st, a, renders, alphas, last_ids, only_tiles);      \
    } while (0)
    switch (kind) {
    case CAM_PERFECT_PINHOLE: GSX_FWD(CAM_PERFECT_PINHOLE); break;
    case CAM_OPENCV_PINHOLE: GSX_FWD(CAM_OPENCV_PINHOLE); break;
    default: GSX_FWD(CAM_OPENCV_FISHEYE); break;
    }
#undef GSX_FWD
    return check_launch("rasterize_to_pixels_from_world_3dgs_fwd");
}

extern "C" const void* gsx_rasterize_fwd_packed_records(const void* fwd_workspace, size_t workspace_bytes, uint32_t C, uint32_t N) {
    if (!fwd_workspace || workspace_bytes < raster_fwd_fast_workspace_bytes(C, N)) return nullptr;
    return (const void*)(((uintptr_t)fwd_workspace + 255) & ~(uintptr_t)255);
}

extern "C" int gsx_rasterize_to_pixels_from_world_3dgs_bwd(
    uint32_t N, int64_t n_isects, const float* means, const float* quats, const float* scales, const float* colors,
    uint32_t channels, const float* opacities, const float* backgrounds, const uint8_t* masks, uint32_t image_width,
    uint32_t image_height, uint32_t tile_size, const gsx_cameras* cams, const gsx_ut_params* ut,
    const int32_t* tile_offsets, const int32_t* flatten_ids, const float* render_alphas, const int32_t* last_ids,
    const float* v_render_colors, const float* v_render_alphas, float* v_means, float* v_quats, float* v_scales,
    float* v_colors, float* v_opacities, void* workspace, size_t workspace_bytes, void* stream) {
    return gsx_rasterize_to_pixels_from_world_3dgs_bwd_packed(N, n_isects, means, quats, scales, colors, channels, opacities, backgrounds, masks,
                                                              image_width, image_height, tile_size, cams, ut, tile_offsets, flatten_ids,
                                                              render_alphas, last_ids, v_render_colors, v_render_alphas, v_means, v_quats, v_scales,
                                                              v_colors, v_opacities, workspace, workspace_bytes, nullptr, stream);
}

extern "C" int gsx_rasterize_to_pixels_from_world_3dgs_bwd_packed(
    uint32_t N, int64_t n_isects, const float* means, const float* quats, const float* scales, const float* colors,
    uint32_t channels, const float* opacities, const float* backgrounds, const uint8_t* masks, uint32_t image_width,
    uint32_t image_height, uint32_t tile_size, const gsx_cameras* cams, const gsx_ut_params* ut,
    const int32_t* tile_offsets, const int32_t* flatten_ids, const float* render_alphas, const int32_t* last_ids,
    const float* v_render_colors, const float* v_render_alphas, float* v_means, float* v_quats, float* v_scales,
    float* v_colors, float* v_opacities, void* workspace, size_t workspace_bytes, const void* packed_records, void* stream) {
    return gsx_rasterize_to_pixels_from_world_3dgs_bwd_guarded(N, n_isects, means, quats, scales, colors, channels, opacities, backgrounds, masks,
                                                               image_width, image_height, tile_size, cams, ut, tile_offsets, flatten_ids,
                                                               render_alphas, last_ids, v_render_colors, v_render_alphas, v_means, v_quats, v_scales,
                                                               v_colors, v_opacities, workspace, workspace_bytes, packed_records, nullptr, 0, stream);
}

// `act` != nullptr: *act_folded tells the caller whether the gather applied the activation Jacobians (else the outputs are the five
// activated-parameter gradients as always and the caller runs gsx_splat_activations_bwd_reg behind it)
static int raster_bwd_impl(
    uint32_t N, int64_t n_isects, const float* means, const float* quats, const float* scales, const float* colors,
    uint32_t channels, const float* opacities, const float* backgrounds, const uint8_t* masks, uint32_t image_width,
    uint32_t image_height, uint32_t tile_size, const gsx_cameras* cams, const gsx_ut_params* ut,
    const int32_t* tile_offsets, const int32_t* flatten_ids, const float* render_alphas, const int32_t* last_ids,
    const float* v_render_colors, const float* v_render_alphas, float* v_means, float* v_quats, float* v_scales,
    float* v_colors, float* v_opacities, void* workspace, size_t workspace_bytes, const void* packed_records,
    const int32_t* lists_status, int64_t n_isects_expected, void* stream, const ActEpilogue* act, bool* act_folded) {
    (void)ut;
    if (act_folded) *act_folded = false;
    RasterArgs a;
    int rc = fill_args(a, N, n_isects, means, quats, scales, colors, channels, opacities, backgrounds, masks, image_width,
                       image_height, tile_size, cams, tile_offsets, flatten_ids, "bwd");
    if (rc != GSX_OK) return rc;
    a.lists_status = lists_status;
    if (lists_status != nullptr && n_isects_expected > 0) a.n_isects_expected = n_isects_expected;   // as the forward: launch decisions from the estimate, not the capacity
    if (!render_alphas || !last_ids || !v_render_colors || !v_means || !v_quats || !v_scales || !v_colors || !v_opacities) {
        set_error("rasterize bwd: null pointer");
        return GSX_ERR_INVALID_ARGUMENT;
    }
    hipStream_t st = (hipStream_t)stream;
    // The five gradient outputs are OVERWRITTEN (upstream accumulates into caller-zeroed tensors: same result for its callers,
    // Rasterization.cpp:196-200): the fast path writes every element, the other paths zero-fill here before their atomics.
    auto zero_outputs = [&]() {
        (void)hipMemsetAsync(v_means, 0, (size_t)N * 3 * 4, st);
        (void)hipMemsetAsync(v_quats, 0, (size_t)N * 4 * 4, st);
        (void)hipMemsetAsync(v_scales, 0, (size_t)N * 3 * 4, st);
        (void)hipMemsetAsync(v_colors, 0, (size_t)a.C * N * 3 * 4, st);
        (void)hipMemsetAsync(v_opacities, 0, (size_t)a.C * N * 4, st);
    };
    if (n_isects == 0 || a.C == 0 || image_width == 0 || image_height == 0) {  // Bwd.cu:434-437
        if (N) zero_outputs();
        return check_launch("rasterize_to_pixels_from_world_3dgs_bwd(empty)");
    }
    const dim3 grid(a.tw, a.th, a.C), block(RB);
    const bool hoist = cams->shutter == GSX_SHUTTER_GLOBAL;
    const int kind = cam_kind(*cams);
    const uint8_t* only_tiles = nullptr;
    bool fast_done = false;
    if (hoist && !force_generic() && (kind != CAM_OPENCV_FISHEYE || ((size_t)a.C * a.tw * a.th <= FAST_FLAG_BYTES && a.lshift == 0))) {
        // the activation epilogue only where the gather's outputs are final: one camera, a pinhole (a fisheye frame may add flagged tiles on top)
        const bool fold = act != nullptr && a.C == 1 && kind != CAM_OPENCV_FISHEYE;
        fast_done = launch_raster_bwd_fast(kind, a, render_alphas, last_ids, v_render_colors, v_render_alphas, v_means, v_quats, v_scales, v_colors,
                                           v_opacities, workspace, workspace_bytes, (const float4*)packed_records, st, &only_tiles, fold ? act : nullptr);
        if (fast_done && fold && act_folded) *act_folded = true;
        if (fast_done && only_tiles == nullptr) return check_launch("rasterize_to_pixels_from_world_3dgs_bwd(fast)");
        // fisheye: the gather kernel has written every output element; the reference-order kernel adds the flagged tiles on top
    }
    if (a.lshift) { set_error("rasterize bwd: lists per 32 x 32 pixels need the fast path (global-shutter pinhole, a workspace of gsx_rasterize_bwd_workspace_bytes(C, N, 4 * n_isects))"); return GSX_ERR_UNSUPPORTED; }
    // Reference-order kernels.  With a workspace and no fast-path results to add to, the per-(tile, Gaussian) gradients travel as
    // chained 64 B records + one gather pass instead of 14 device-scope float atomics each (rolling shutter, forced generic path).
    float4* grad_rec = nullptr;
    int32_t* grad_head = nullptr;
    const size_t rec_bytes = ((size_t)n_isects * 64 + 255) / 256 * 256;
    if (!fast_done && workspace != nullptr && workspace_bytes >= rec_bytes + (size_t)a.C * N * 4) {
        grad_rec = (float4*)workspace;
        grad_head = (int32_t*)((char*)workspace + rec_bytes);
        (void)hipMemsetAsync(grad_head, 0xFF, (size_t)a.C * N * 4, st);
    } else if (!fast_done) {
        zero_outputs();
    }
#define GSX_BWD(KIND)                                                                                                  \
    do {                                                                                                               \
        if (hoist) hipLaunchKernelGGL(HIP_KERNEL_NAME(raster_bwd_kernel<KIND, true>), grid, block, 0, st, a, render_alphas, last_ids, v_render_colors, v_render_alphas, v_means, v_quats, v_scales, v_colors, v_opacities, only_tiles, grad_rec, grad_head); \
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(raster_bwd_kernel<KIND, false>), grid, block, 0, st, a, render_alphas, last_ids, v_render_colors, v_render_alphas, v_means, v_quats, v_scales, v_colors, v_opacities, only_tiles, grad_rec, grad_head);      \
    } while (0)
    switch (kind) {
    case CAM_PERFECT_PINHOLE: GSX_BWD(CAM_PERFECT_PINHOLE); break;
    case CAM_OPENCV_PINHOLE: GSX_BWD(CAM_OPENCV_PINHOLE); break;
    default: GSX_BWD(CAM_OPENCV_FISHEYE); break;
    }
#undef GSX_BWD
    if (grad_rec != nullptr)
        hipLaunchKernelGGL(gsx_bwd_gather_grads_kernel, dim3((N + 255u) / 256u), dim3(256), 0, st, a.C, N, (const float4*)grad_rec,
                           (const int32_t*)grad_head, v_means, v_quats, v_scales, v_colors, v_opacities);
    return check_launch("rasterize_to_pixels_from_world_3dgs_bwd");
}

extern "C" int gsx_rasterize_to_pixels_from_world_3dgs_bwd_guarded(
    uint32_t N, int64_t n_isects, const float* means, const float* quats, const float* scales, const float* colors,
    uint32_t channels, const float* opacities, const float* backgrounds, const uint8_t* masks, uint32_t image_width,
    uint32_t image_height, uint32_t tile_size, const gsx_cameras* cams, const gsx_ut_params* ut,
    const int32_t* tile_offsets, const int32_t* flatten_ids, const float* render_alphas, const int32_t* last_ids,
    const float* v_render_colors, const float* v_render_alphas, float* v_means, float* v_quats, float* v_scales,
    float* v_colors, float* v_opacities, void* workspace, size_t workspace_bytes, const void* packed_records,
    const int32_t* lists_status, int64_t n_isects_expected, void* stream) {
    return raster_bwd_impl(N, n_isects, means, quats, scales, colors, channels, opacities, backgrounds, masks, image_width, image_height, tile_size, cams, ut,
                           tile_offsets, flatten_ids, render_alphas, last_ids, v_render_colors, v_render_alphas, v_means, v_quats, v_scales, v_colors,
                           v_opacities, workspace, workspace_bytes, packed_records, lists_status, n_isects_expected, stream, nullptr, nullptr);
}

// ABI 7: the blend backward of ONE camera through to the RAW SplatData parameters (scaling_raw / rotation_raw / opacity_raw): on the fast path the
// gather kernel applies the activation Jacobians where it holds the activated-parameter gradients in registers (ActEpilogue), elsewhere
// gsx_splat_activations_bwd_reg runs behind the blend backward — the same values either way.  v_quats / v_scales / v_opacities are scratch
// (written only on the second route).  (means, quats, scales, opacities must be the activations of the raw tensors: splat_data.cpp:267-286.)
extern "C" int gsx_rasterize_to_pixels_from_world_3dgs_bwd_act(
    uint32_t N, int64_t n_isects, const float* means, const float* quats, const float* scales, const float* colors,
    uint32_t channels, const float* opacities, const float* backgrounds, const uint8_t* masks, uint32_t image_width,
    uint32_t image_height, uint32_t tile_size, const gsx_cameras* cams, const gsx_ut_params* ut,
    const int32_t* tile_offsets, const int32_t* flatten_ids, const float* render_alphas, const int32_t* last_ids,
    const float* v_render_colors, const float* v_render_alphas, float* v_means, float* v_quats, float* v_scales,
    float* v_colors, float* v_opacities, void* workspace, size_t workspace_bytes, const void* packed_records,
    const int32_t* lists_status, int64_t n_isects_expected, const float* scaling_raw, const float* rotation_raw, const float* opacity_raw,
    float* v_scaling_raw, float* v_rotation_raw, float* v_opacity_raw, float scale_reg_per_element, float opacity_reg_per_element, void* stream) {
    if (!scaling_raw || !rotation_raw || !opacity_raw || !v_scaling_raw || !v_rotation_raw || !v_opacity_raw) {
        set_error("rasterize bwd (act): null pointer");
        return GSX_ERR_INVALID_ARGUMENT;
    }
    if (cams == nullptr || cams->C != 1) { set_error("rasterize bwd (act): one camera only (opacities are per camera)"); return GSX_ERR_UNSUPPORTED; }
    const ActEpilogue act{rotation_raw, v_scaling_raw, v_rotation_raw, v_opacity_raw, scale_reg_per_element, opacity_reg_per_element};
    bool folded = false;
    const int rc = raster_bwd_impl(N, n_isects, means, quats, scales, colors, channels, opacities, backgrounds, masks, image_width, image_height, tile_size, cams,
                                   ut, tile_offsets, flatten_ids, render_alphas, last_ids, v_render_colors, v_render_alphas, v_means, v_quats, v_scales,
                                   v_colors, v_opacities, workspace, workspace_bytes, packed_records, lists_status, n_isects_expected, stream, &act, &folded);
    if (rc != GSX_OK || folded) return rc;
    return gsx_splat_activations_bwd_reg(N, scaling_raw, rotation_raw, opacity_raw, v_scales, v_quats, v_opacities, v_scaling_raw, v_rotation_raw,
                                         v_opacity_raw, scale_reg_per_element, opacity_reg_per_element, stream);
}

extern "C" size_t gsx_rasterize_fwd_workspace_bytes(uint32_t C, uint32_t N) { return raster_fwd_fast_workspace_bytes(C, N); }

extern "C" size_t gsx_rasterize_bwd_workspace_bytes(uint32_t C, uint32_t N, int64_t n_isects) {
    return raster_bwd_fast_workspace_bytes(C, N, n_isects);
}
