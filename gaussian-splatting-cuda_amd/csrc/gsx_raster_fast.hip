// gsx_raster_fast.hip — the MI355X fast path of the world-space blend (global shutter, pinhole cameras
// with or without OpenCV distortion).  Same operator semantics as gsx_raster.hip (reference:
// gsplat/RasterizeToPixelsFromWorld3DGSFwd.cu:19-279, ...Bwd.cu:16-373), different arithmetic route.
//
// 1. Algebra.  For a global shutter every pixel ray starts at the camera centre c, so with
//    A = M Rc (M = diag(1/s) R^T, Rc = camera->world rotation), m = Rc^-1 (mu - c) the camera-space centre (the exact inverse, not the transpose: gsx_record.hpp make_cam_frame),
//    p = (u, v, 1) the pixel's undistorted normalised coordinates and g = M (c - mu) = -A m:
//        grayDist = |(A p) x g|^2 / |A p|^2                       (scale of the ray direction cancels)
//        (A p) x g = -(A p) x (A m) = -cof(A) (p x m)             (cof(A) columns = a_i x a_j)
//        p x m     = m_z (dv, -du, du v0 - dv u0),   (u0,v0) = (m_x,m_y)/m_z,  (du,dv) = (u-u0, v-v0)
//    => (A p) x g = du B0 + dv B1 with two per-Gaussian 3-vectors.  The reference evaluates the cross
//    product of a unit vector with g whose norm is depth/scale ~ 1e2..1e4 and loses that many digits to
//    cancellation; here the only subtraction is u - u0.  |du B0 + dv B1|^2 is evaluated through the 2x2
//    triangular factor L of [B0 B1] (two squares, no cancellation); |A p|^2 is a convex quadratic in
//    (du,dv) normalised to 1 at the centre.  16 VALU per (pixel, Gaussian) instead of ~35, and closer to
//    the float64 truth than the reference's own fp32 order (tests/test_gpu_ops.py measures both).
// 2. Sub-tile culling.  Each wave owns an 8x8 quadrant.  alpha >= 1/255 needs
//    |L d|^2 <= log2(255 o) * den(d); den is convex, so its maximum over the tile is at a corner and the
//    ellipse |L d|^2 <= log2(255 o) * max den is a conservative footprint, tested EXACTLY against the rectangle of the
//    wave's pixel centres (footprint_hits: minimum of the form over the rectangle).  Every wave tests 64 staged
//    Gaussians at a time (one per lane) against its quadrant, ballots the survivors and walks only
//    the set bits, front to back.  A skipped Gaussian has alpha < 1/255 on all 64 pixels, so results are
//    unchanged (the reference `continue`s on exactly those pairs, Fwd.cu:240).
// 3. Staging.  pack_records_kernel turns every (camera, Gaussian) into ONE 64 B record (centre, factor L, log2 opacity, the
//    normalised denominator quadratic, colour) once per launch; a tile gathers one cache line per intersection.  Chunks of
//    128 Gaussians, records in LDS as AoS float4 x 4 (four wave-uniform ds_read_b128 in the pixel loop) plus a separate
//    float4 cull plane (u0, v0, rad2, k2: conflict-free per-lane reads), double buffered in the forward: the flatten ids of
//    chunk b+1 are in flight while chunk b is composited; one barrier per chunk.  The backward of the same inputs can take
//    the forward's packed records back (gsx_rasterize_..._bwd_packed).
// 4. Dispatch.  1-D grid, XCD-aware: block b runs on XCD b % 8, so each XCD is given a contiguous band of
//    tiles and neighbouring tiles (which share Gaussians) hit the same 4 MiB L2.
#include "gsx_raster_common.hpp"
#include "gsx_record.hpp"

#include <cstdlib>
#include <string>

namespace gsx {

#ifdef GSX_STATS
// debug build only (tools/fwd_stats.py): counters for tuning the culling / early-exit behaviour
__device__ unsigned long long g_stats[16];
#define GSX_STAT_ADD(i, v) do { if (lane == 0) atomicAdd(&g_stats[i], (unsigned long long)(v)); } while (0)
#else
#define GSX_STAT_ADD(i, v) do { } while (0)
#endif
#ifdef GSX_CLOCKS
// debug build only (tools/blend_clock.py, -DGSX_CLOCKS; no other counters, so the kernels run at their production speed)
// the shader clock a blend kernel really runs at (round 5): thread 0 of every block brackets its block with s_memtime (shader cycles:
// MI355X_MICROARCH.md) and s_memrealtime (constant 100 MHz); g_clk[3 k ..] = {sum of cycles, sum of 10 ns ticks, blocks} of kernel k
// (0 one-list forward, 1 four-list forward, 2 Gaussian-major backward, 3 pixel-major backward).  tools/blend_clock.py
__device__ unsigned long long g_clk[12];
struct BlockClock {
    unsigned long long t0, r0; int k; bool on;
    __device__ BlockClock(int k_) : k(k_), on(threadIdx.x == 0) { if (on) { t0 = __builtin_readcyclecounter(); r0 = wall_clock64(); } }
    __device__ ~BlockClock() { if (on) { atomicAdd(&g_clk[3 * k], __builtin_readcyclecounter() - t0); atomicAdd(&g_clk[3 * k + 1], wall_clock64() - r0); atomicAdd(&g_clk[3 * k + 2], 1ull); } }
};
#define GSX_BLOCK_CLOCK(k) BlockClock gsx_block_clock(k)
#else
#define GSX_BLOCK_CLOCK(k) do { } while (0)
#endif

// minimum waves per SIMD requested from the register allocator (__launch_bounds__ 2nd argument)
#ifndef GSX_FWD_WAVES
#define GSX_FWD_WAVES 8
#endif
#ifndef GSX_BWD_WAVES
#define GSX_BWD_WAVES 4
#endif

#ifndef GSX_FCH
#define GSX_FCH 128
#endif
constexpr int FCH = GSX_FCH;  // Gaussians per forward chunk (double buffered)
// ---- packed per-(camera, Gaussian) records ---------------------------------------------------------------
// The tile kernels need, per (tile, Gaussian), data that lives in five separate arrays (means 12 B, quats 16 B,
// scales 12 B, opacities 4 B, colours 12 B): five cache lines gathered for 60 useful bytes, and ~300 VALU to turn
// them into the 15 camera-space coefficients — all of which depend on (camera, Gaussian) only, not on the tile.
// pack_records_kernel does that once per (camera, Gaussian) into ONE 64 B line; staging a tile then gathers a
// single line per Gaussian and only adds the tile-dependent footprint (rad2, k2).
//   p0 = (u0, v0, l00, l01)  p1 = (l11, lo, d1, d2)  p2 = (d3, d4, d5, red)  p3 = (green, blue, -, -)
// `bad` (fisheye only): 1 where the Gaussian has no usable (u0, v0) chart — it sits behind the camera plane or more than
// atan(8) = 83 degrees off the optical axis, where u0 = m_x / m_z loses its digits.  Such a Gaussian is visible in a wide fisheye;
// the tiles that list one are rendered by the reference-order kernels (tile_flag_kernel below).
__global__ __launch_bounds__(256) void pack_records_kernel(RasterArgs a, float4* __restrict__ packed, int32_t* __restrict__ heads,
                                                           uint8_t* __restrict__ bad) {
    const uint32_t n = blockIdx.x * 256u + threadIdx.x, c = blockIdx.y;
    if (n >= a.N) return;
    const size_t g = (size_t)c * a.N + n;
    if (heads) {  // backward: empty moment-record chains (NSUB planes), mode word = chains
        const size_t cn = (size_t)a.C * a.N;
#pragma unroll
        for (int k = 0; k < NSUB; ++k) heads[(size_t)k * cn + g] = -1;
        if (g == 0) reinterpret_cast<uint32_t*>(heads + (align256(cn * 4 * NSUB) >> 2))[1] = 0u;
    }
    RawG raw;
    raw.g = (int32_t)g;
    raw.mu = {a.means[(size_t)n * 3], a.means[(size_t)n * 3 + 1], a.means[(size_t)n * 3 + 2]};
    raw.q = reinterpret_cast<const float4*>(a.quats)[n];
    raw.sc = {a.scales[(size_t)n * 3], a.scales[(size_t)n * 3 + 1], a.scales[(size_t)n * 3 + 2]};
    raw.opac = a.opacities[g];
    raw.rgb = {a.colors[g * 3], a.colors[g * 3 + 1], a.colors[g * 3 + 2]};
    const ShutterPoses sp(a.cams.viewmats0 + c * 16, nullptr);
    const CamFrame cf = make_cam_frame(sp);
    float4* o = packed + g * 4;
    store_packed_record(raw, cf, o);
    if (bad) {
        const f3 dm = raw.mu - cf.c;
        const float mz = cf.Rci[2][0] * dm.x + cf.Rci[2][1] * dm.y + cf.Rci[2][2] * dm.z;
        const float4 p0 = o[0];
        bad[g] = (mz > 0.f && fabsf(p0.x) <= 8.f && fabsf(p0.y) <= 8.f) ? 0 : 1;   // (NaN compares false: bad)
    }
}

// fisheye: flags[camera, tile] = 1 when the tile's list holds a Gaussian without a usable chart.  One wave per tile.
__global__ __launch_bounds__(256) void tile_flag_kernel(RasterArgs a, const uint8_t* __restrict__ bad, uint8_t* __restrict__ flags) {
    const uint32_t n_tiles = a.tw * a.th, t = blockIdx.x * 4u + (threadIdx.x >> 6), cid = blockIdx.y, lane = threadIdx.x & 63u;
    if (t >= n_tiles) return;
    const int32_t* toff = a.tile_offsets + (size_t)cid * n_tiles;
    bool lists_ok;
    const int32_t total = lists_total(a, lists_ok);
    const int32_t lo = toff[t], hi = !lists_ok ? lo : ((cid == a.C - 1 && t == n_tiles - 1) ? total : toff[t + 1]);
    bool any = false;
    for (int32_t i = lo + (int32_t)lane; i < hi; i += 64) any = any || bad[a.flatten_ids[i]] != 0;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(any);
    if (lane == 0) flags[(size_t)cid * n_tiles + t] = m != 0ull ? 1 : 0;
}

// alpha of one (pixel, Gaussian) pair: 16 VALU.  Record layout (== the packed 64 B record):
//   r0 = (u0, v0, l00, l01)  r1 = (l11, lo, d1, d2)  r2 = (d3, d4, d5, red)  r3 = (green, blue, -, -)
// Returns alpha; (x0, x1) = the whitened pixel offsets L (du, dv) (N = x0^2 + x1^2: the backward takes its moments in them, gsx_record.hpp),
// num2 = 0.5 log2(e) * grayDist * den' (scaled numerator), rden = 1/den'.
GSX_DEV float fast_alpha(float u, float v, float4 r0, float4 r1, float4 r2, float& x0, float& x1, float& num2, float& rden) {
    const float du = u - r0.x, dv = v - r0.y;
    const float t0 = fmaf(r0.w, dv, r0.z * du);
    const float t1 = r1.x * dv;
    x0 = t0; x1 = t1;
    num2 = fmaf(t0, t0, t1 * t1);
    const float den = fmaf(du, fmaf(r2.x, du, fmaf(r2.y, dv, r1.z)), fmaf(dv, fmaf(r2.z, dv, r1.w), 1.f));
    rden = __builtin_amdgcn_rcpf(den);
    return fminf(0.999f, __builtin_amdgcn_exp2f(fmaf(-num2, rden, r1.y)));
}

// The same for a ray given as an unnormalised direction (u, v, w) instead of (u, v, 1) — fisheye, where rays reach and pass 90 degrees
// off axis (w <= 0).  With du' = u - w u0, dv' = v - w v0:  (A d) x g = du' B0 + dv' B1 and A d = w h + a0 du' + a1 dv', so the numerator
// keeps its form and the denominator becomes w (w + d1 du' + d2 dv') + d3 du'^2 + d4 du' dv' + d5 dv'^2; the ratio does not depend
// on the length of (u, v, w).  19 VALU.
GSX_DEV float fast_alpha_ray(float u, float v, float w, float ww, float4 r0, float4 r1, float4 r2, float& x0, float& x1, float& num2, float& rden) {
    const float du = fmaf(-w, r0.x, u), dv = fmaf(-w, r0.y, v);
    const float t0 = fmaf(r0.w, dv, r0.z * du);
    const float t1 = r1.x * dv;
    x0 = t0; x1 = t1;
    num2 = fmaf(t0, t0, t1 * t1);
    const float den = fmaf(du, fmaf(r2.x, du, fmaf(r2.y, dv, r1.z * w)), fmaf(dv, fmaf(r2.z, dv, r1.w * w), ww));
    rden = __builtin_amdgcn_rcpf(den);
    return fminf(0.999f, __builtin_amdgcn_exp2f(fmaf(-num2, rden, r1.y)));
}

// Conservative footprint of a record for the tile bounds tb (see header comment, item 2), as an ellipse around (u0, v0):
//     alpha >= 1/255   =>   N(d) = (l00 du + l01 dv)^2 + (l11 dv)^2  <=  tau2 den(d)  <=  rad2 := tau2 * max over the tile of den
// (den is convex: its maximum over the tile sits at a corner).  rad2 < 0: never visible (opacity <= 1/255); NaN / +inf: degenerate
// factor, never culled.  k2 = -l00 l01 / (l01^2 + l11^2) is the slope of the line of minima of N along dv (footprint_hits).
GSX_DEV void footprint(float4 r0, float4 r1, float4 r2, const float tb[4], float& rad2, float& k2) {
    const float tau2 = r1.y + LOG2_255;
    rad2 = -1.f;
    k2 = -(r0.z * r0.w) * __builtin_amdgcn_rcpf(fmaf(r0.w, r0.w, r1.x * r1.x));
    if (tau2 > 0.f) {
        float dmax = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float du = ((k & 1) ? tb[1] : tb[0]) - r0.x, dv = ((k & 2) ? tb[3] : tb[2]) - r0.y;
            dmax = fmaxf(dmax, 1.f + du * (r1.z + r2.x * du + r2.y * dv) + dv * (r1.w + r2.z * dv));
        }
        rad2 = fmaf(tau2 * dmax, 1.0021f, 1e-6f);
    }
}

// Does the footprint ellipse {N(d) <= rad2} of a staged Gaussian reach the rectangle [b0, b1] x [b2, b3] of pixel centres?  EXACT: the
// minimum of the convex form N over the rectangle, which lies on one of the two sides facing the centre (or is 0 inside): with
// (xc, yc) the rectangle's point closest to the centre per axis, the minimum over the side dv = yc is at l00 du = clamp(-l01 yc) and
// the minimum over the side du = xc at dv = clamp(k2 xc).  The bounding-box test this replaces let an ellipse through whenever its BOX
// touched the rectangle: 23 % of the (Gaussian, 4x4 block) pairs and 15 % of the (Gaussian, 8x8 quadrant) pairs of S-1M, all of them
// evaluated for nothing (thin ellipses at an angle).  !(min > rad2): a NaN anywhere means "not culled".
GSX_DEV bool footprint_hits(float4 c, float l00, float l01, float l11, float b0, float b1, float b2, float b3) {
    const float xa = b0 - c.x, xb = b1 - c.x, ya = b2 - c.y, yb = b3 - c.y;
    const float xc = __builtin_amdgcn_fmed3f(0.f, xa, xb), yc = __builtin_amdgcn_fmed3f(0.f, ya, yb);
    const float m = l01 * yc;
    const float t = __builtin_amdgcn_fmed3f(-m, l00 * xa, l00 * xb) + m, e1 = l11 * yc;
    const float n1 = fmaf(t, t, e1 * e1);
    const float ys = __builtin_amdgcn_fmed3f(c.w * xc, ya, yb);
    const float t2 = fmaf(l01, ys, l00 * xc), e2 = l11 * ys;
    const float n2 = fmaf(t2, t2, e2 * e2);
    return !(fminf(n1, n2) > c.z);
}

// The same test for the four blocks of a 2 x 2 arrangement whose rectangles are products of two u-ranges (xr[0], xr[1]) and two v-ranges
// (yr[0], yr[1]) — the four 4x4 blocks of a wave's quadrant under a perfect pinhole (block sb = 2 sy + sx).  The 1-D pieces (clamped
// offsets, their products with the factor) are formed once per range: ~60 instead of 4 x 23 VALU.  Bit sb of the result = block sb is reached.
GSX_DEV uint32_t footprint_hits_2x2(float4 c, float l00, float l01, float l11, const float (&xr)[2][2], const float (&yr)[2][2]) {
    float xa[2], xb[2], xc[2], lxa[2], lxb[2], lxc[2], kxc[2], ya[2], yb[2], m[2], e1s[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        xa[k] = xr[k][0] - c.x; xb[k] = xr[k][1] - c.x;
        xc[k] = __builtin_amdgcn_fmed3f(0.f, xa[k], xb[k]);
        lxa[k] = l00 * xa[k]; lxb[k] = l00 * xb[k]; lxc[k] = l00 * xc[k]; kxc[k] = c.w * xc[k];
        ya[k] = yr[k][0] - c.y; yb[k] = yr[k][1] - c.y;
        const float yc = __builtin_amdgcn_fmed3f(0.f, ya[k], yb[k]);
        m[k] = l01 * yc;
        const float e1 = l11 * yc;
        e1s[k] = e1 * e1;
    }
    uint32_t hits = 0u;
#pragma unroll
    for (int sb = 0; sb < 4; ++sb) {
        const int sx = sb & 1, sy = sb >> 1;
        const float t = __builtin_amdgcn_fmed3f(-m[sy], lxa[sx], lxb[sx]) + m[sy];
        const float n1 = fmaf(t, t, e1s[sy]);
        const float ys = __builtin_amdgcn_fmed3f(kxc[sx], ya[sy], yb[sy]);
        const float t2 = fmaf(l01, ys, lxc[sx]), e2 = l11 * ys;
        const float n2 = fmaf(t2, t2, e2 * e2);
        hits |= (!(fminf(n1, n2) > c.z)) ? (1u << sb) : 0u;
    }
    return hits;
}

// one staged Gaussian: the 64 B record (AoS, read at a wave-uniform index with one base address) + the cull plane entry
struct StagedRec { float4 r0, r1, r2, r3, cull; };   // cull = (u0, v0, rad2, k2): footprint()

GSX_DEV void stage_one(const RasterArgs& a, const float tb[4], int32_t g, StagedRec& o, uint32_t tile_x, uint32_t tile_y) {
    const float4* p = a.packed + (size_t)g * 4;
    o.r0 = p[0]; o.r1 = p[1]; o.r2 = p[2]; o.r3 = p[3];
    float rad2, k2;
    footprint(o.r0, o.r1, o.r2, tb, rad2, k2);
    // lists per 32 x 32 pixels: the parent's list also names Gaussians whose rectangle of 16-pixel tiles (IntersectTile.cu:65-76) does not
    // contain THIS tile — the reference never composites those here, whatever their alpha: an empty footprint drops them at staging
    if (a.lshift != 0u && a.rect_filter != 0u && !rect_has_tile(o.r3, tile_x, tile_y)) rad2 = -1.f;
    o.cull = make_float4(o.r0.x, o.r0.y, rad2, k2);
}

// per-thread pixel set-up shared by forward and backward: undistorted normalised coordinates (u,v)
template <int KIND>
GSX_DEV bool pixel_uv(const Camera<KIND>& cam, uint32_t i, uint32_t j, float& u, float& v) {
    const float px = (float)j + 0.5f, py = (float)i + 0.5f;
    if (KIND == CAM_PERFECT_PINHOLE) {
        u = (px - cam.cx) / cam.fx;
        v = (py - cam.cy) / cam.fy;
        return true;
    }
    f3 d;
    const bool ok = cam.unproject(f2{px, py}, d);
    u = d.x / d.z; v = d.y / d.z;
    return ok;
}

// fisheye: the pixel's ray as a unit vector (u, v, w), w <= 0 at and beyond 90 degrees; pinholes: (u, v, 1)
template <int KIND>
GSX_DEV bool pixel_ray(const Camera<KIND>& cam, uint32_t i, uint32_t j, float& u, float& v, float& w) {
    if (KIND != CAM_OPENCV_FISHEYE) { w = 1.f; return pixel_uv(cam, i, j, u, v); }
    f3 d;
    const bool ok = cam.unproject(f2{(float)j + 0.5f, (float)i + 0.5f}, d);
    u = d.x; v = d.y; w = d.z;
    return ok;
}

// tile bounds in (u,v): min/max over the valid pixels of each wave -> LDS -> whole tile
// (`wide`: a fisheye pixel whose ray is near or beyond 90 degrees has no (u, v): its wave's and the tile's bounds become infinite,
// which switches the footprint culling off for them)
GSX_DEV void uv_bounds(bool valid, float u, float v, uint32_t wave, uint32_t lane, float (*s_bounds)[4], float wb[4],
                       float tb[4], bool wide = false) {
    float umin = wide ? -INFINITY : (valid ? u : INFINITY), umax = wide ? INFINITY : (valid ? u : -INFINITY);
    float vmin = wide ? -INFINITY : (valid ? v : INFINITY), vmax = wide ? INFINITY : (valid ? v : -INFINITY);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        umin = fminf(umin, __shfl_xor(umin, o)); umax = fmaxf(umax, __shfl_xor(umax, o));
        vmin = fminf(vmin, __shfl_xor(vmin, o)); vmax = fmaxf(vmax, __shfl_xor(vmax, o));
    }
    wb[0] = umin; wb[1] = umax; wb[2] = vmin; wb[3] = vmax;
    if (lane == 0) { s_bounds[wave][0] = umin; s_bounds[wave][1] = umax; s_bounds[wave][2] = vmin; s_bounds[wave][3] = vmax; }
    __syncthreads();
    tb[0] = fminf(fminf(s_bounds[0][0], s_bounds[1][0]), fminf(s_bounds[2][0], s_bounds[3][0]));
    tb[1] = fmaxf(fmaxf(s_bounds[0][1], s_bounds[1][1]), fmaxf(s_bounds[2][1], s_bounds[3][1]));
    tb[2] = fminf(fminf(s_bounds[0][2], s_bounds[1][2]), fminf(s_bounds[2][2], s_bounds[3][2]));
    tb[3] = fmaxf(fmaxf(s_bounds[0][3], s_bounds[1][3]), fmaxf(s_bounds[2][3], s_bounds[3][3]));
}

// XCD-aware 1-D grid -> tile id (block b is dispatched to XCD b % 8)
GSX_DEV bool swizzled_tile(uint32_t b, uint32_t n_tiles, uint32_t& tile_id) {
    const uint32_t per = (n_tiles + 7u) / 8u;
    tile_id = (b & 7u) * per + (b >> 3);
    return (b >> 3) < per && tile_id < n_tiles;
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(RB, GSX_FWD_WAVES) void raster_fwd_fast_kernel(RasterArgs a, float* __restrict__ render_colors,
                                                             float* __restrict__ render_alphas,
                                                             int32_t* __restrict__ last_ids) {
    // AoS records, double buffered, pitch FIVE float4 (80 B): [0..3] the record, [4] = (rad2, k2, -, -) of the footprint test.  The per-lane
    // reads of the culling pass have the pitch as their stride: 80 B spreads a 16-lane group of a ds_read_b128 over all 64 banks (64 B: four-way
    // conflicts); the step loop reads one record wave-uniformly either way.  (Layout shared with raster_fwd_quad_kernel.)
    __shared__ float4 s_rec[2][FCH][5];
    __shared__ float s_bounds[4][4];
    __shared__ int s_wdone[2][4];
    GSX_BLOCK_CLOCK(0);
    const uint32_t cid = blockIdx.y;
    uint32_t tile_id;
    if (!swizzled_tile(blockIdx.x, a.tw * a.th, tile_id)) return;
    const uint32_t tile_y = tile_id / a.tw, tile_x = tile_id - tile_y * a.tw;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
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
    if (KIND == CAM_OPENCV_FISHEYE && a.tile_flags != nullptr && a.tile_flags[(size_t)cid * a.th * a.tw + tile_id]) return;  // generic kernel's tile
    const Camera<KIND> cam(a.cams, cid, a.W, a.H);
    float u, v, w;
    const bool ray_ok = pixel_ray(cam, i, j, u, v, w);
    bool done = !inside || !ray_ok;
    float wb[4], tb[4];
    if (KIND == CAM_OPENCV_FISHEYE) {
        const bool wide = !done && w < 0.05f;
        const float iw = 1.f / fmaxf(w, 0.05f);
        uv_bounds(!done, u * iw, v * iw, wave, lane, s_bounds, wb, tb, wide);
    } else {
        uv_bounds(!done, u, v, wave, lane, s_bounds, wb, tb);
    }
    const float ww = w * w;
    const bool no_cull = KIND == CAM_OPENCV_FISHEYE && !(tb[0] > -INFINITY);

    int32_t range_start, range_end;
    tile_list_range(a, cid, tile_x, tile_y, range_start, range_end);
    const int32_t n_chunks = (range_end - range_start + FCH - 1) / FCH;

    // Compositing state.  The alpha clamp min(0.999, .) is folded into the exponential: the loop works with alpha' = alpha / 0.999
    // = clamp01(exp2(lo' - ...)) (lo' = lo - log2 0.999 and colours scaled by 0.999 at staging time; the [0,1] clamp is an output
    // modifier of v_exp_f32, not an instruction).  A pixel that is finished carries the threshold +inf, so "skip below 1/255" and
    // "done" are ONE comparison; the T <= 1e-4 stop (rare: a pixel stops once) is handled in a wave-uniform side branch.
    constexpr float K999 = 0.999f, LOG2_K999 = -0.0014434168696687174f, THR = (1.f / 255.f) / 0.999f;
    float T = 1.f;
    uint32_t cur_idx = 0;
    float out_r = 0.f, out_g = 0.f, out_b = 0.f;
    float thr = done ? INFINITY : THR;
    bool wave_done = __builtin_amdgcn_ballot_w64(!done) == 0ull;
    int32_t g_pre = 0;  // the flatten id of the next chunk is prefetched, its 64 B packed record is gathered at staging time
    bool have = (int32_t)tid < FCH && range_start + (int32_t)tid < range_end;
    if (have) g_pre = a.flatten_ids[range_start + (int32_t)tid];
    for (int32_t b = 0; b < n_chunks; ++b) {
        const int buf = b & 1;
        const int32_t chunk_start = range_start + FCH * b;
        if (have) {
            StagedRec sr;
            stage_one(a, tb, g_pre, sr, tile_x, tile_y);
            if (no_cull) sr.cull.z = INFINITY;
            sr.r1.y -= LOG2_K999; sr.r2.w *= K999; sr.r3.x *= K999; sr.r3.y *= K999;
            s_rec[buf][tid][0] = sr.r0; s_rec[buf][tid][1] = sr.r1; s_rec[buf][tid][2] = sr.r2; s_rec[buf][tid][3] = sr.r3;
            s_rec[buf][tid][4] = make_float4(sr.cull.z, sr.cull.w, 0.f, 0.f);
        }
        if (lane == 0) s_wdone[buf][wave] = wave_done ? 1 : 0;
        __syncthreads();
        if (s_wdone[buf][0] & s_wdone[buf][1] & s_wdone[buf][2] & s_wdone[buf][3]) break;  // Fwd.cu:188-190
        have = (b + 1 < n_chunks) && (int32_t)tid < FCH && (chunk_start + FCH + (int32_t)tid < range_end);
        if (have) g_pre = a.flatten_ids[chunk_start + FCH + (int32_t)tid];  // in flight during the pixel loop
        if (wave_done) continue;
        const int32_t chunk_size = min(FCH, range_end - chunk_start);
        for (int32_t sub = 0; sub < chunk_size && !wave_done; sub += 64) {
            // one candidate Gaussian per lane: does its footprint touch this wave's quadrant?
            bool hit = false;
            if (sub + (int32_t)lane < chunk_size) {
                const float4 q0 = s_rec[buf][sub + lane][0], q1 = s_rec[buf][sub + lane][1], q4 = s_rec[buf][sub + lane][4];   // (u0, v0, l00, l01), (l11, ..), (rad2, k2)
                const float4 c = make_float4(q0.x, q0.y, q4.x, q4.y);
                hit = footprint_hits(c, q0.z, q0.w, q1.x, wb[0], wb[1], wb[2], wb[3]);
            }
            unsigned long long todo = __builtin_amdgcn_ballot_w64(hit);
            GSX_STAT_ADD(1, min(64, chunk_size - sub));
            GSX_STAT_ADD(0, __popcll(todo));
            while (todo) {
                const int t = sub + __builtin_ctzll(todo);
                todo &= todo - 1ull;
                const float4* rp = s_rec[buf][t];
                const float4 r0 = rp[0], r1 = rp[1], r2 = rp[2], r3 = rp[3];
                const float du = KIND == CAM_OPENCV_FISHEYE ? fmaf(-w, r0.x, u) : u - r0.x;   // fisheye: unnormalised ray (u, v, w), see fast_alpha_ray
                const float dv = KIND == CAM_OPENCV_FISHEYE ? fmaf(-w, r0.y, v) : v - r0.y;
                const float t0 = fmaf(r0.w, dv, r0.z * du);
                const float t1 = r1.x * dv;
                const float num2 = fmaf(t0, t0, t1 * t1);
                const float den = KIND == CAM_OPENCV_FISHEYE
                                      ? fmaf(du, fmaf(r2.x, du, fmaf(r2.y, dv, r1.z * w)), fmaf(dv, fmaf(r2.z, dv, r1.w * w), ww))
                                      : fmaf(du, fmaf(r2.x, du, fmaf(r2.y, dv, r1.z)), fmaf(dv, fmaf(r2.z, dv, r1.w), 1.f));
                const float ap = __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(fmaf(-num2, __builtin_amdgcn_rcpf(den), r1.y)), 0.f, 1.f);
                bool take = ap >= thr;                       // alpha >= 1/255 and the pixel is not finished (Fwd.cu:240)
                float w = take ? ap * T : 0.f;               // alpha T / 0.999
                T = fmaf(-K999, w, T);                       // T (1 - alpha), in place
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(T <= 1e-4f) != 0ull, 0)) {
                    // some pixel stops here: it does NOT take this Gaussian (Fwd.cu:245-248).  Its transmittance is put back
                    // (T + alpha T: equal to the old value to one rounding; only 1 - T of a finished pixel is ever read again).
                    const bool stop = T <= 1e-4f;
                    take = take && !stop;
                    T = stop ? fmaf(K999, w, T) : T;
                    w = stop ? 0.f : w;
                    thr = stop ? INFINITY : thr;
                }
                out_r = fmaf(r2.w, w, out_r); out_g = fmaf(r3.x, w, out_g); out_b = fmaf(r3.y, w, out_b);
                cur_idx = take ? (uint32_t)(chunk_start + t) : cur_idx;
#ifdef GSX_STATS
                { const unsigned long long c = __builtin_amdgcn_ballot_w64(take); GSX_STAT_ADD(2, c != 0ull); GSX_STAT_ADD(3, __popcll(c)); }
#endif
            }
            // all 64 pixels finished: tested once per 64 candidates, not per Gaussian
            if (__builtin_amdgcn_ballot_w64(thr < INFINITY) == 0ull) wave_done = true;
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

// ---- forward, four lists per wave ("quad" variant) -------------------------------------------------------------------------------
// Same arithmetic per (pixel, Gaussian) pair and the same order per pixel as raster_fwd_fast_kernel (bit-identical output), but the wave
// no longer walks ONE list of the Gaussians that reach its 8x8 quadrant: its four DPP rows own the four 4x4 pixel blocks of the quadrant
// and every row walks the list of the Gaussians that reach ITS block — four different Gaussians per wave instruction.  At S-1M a
// Gaussian that reaches a quadrant reaches 2.4 of its 4 blocks on average, so the longest of the four lists is ~0.67 of the quadrant's
// list: a third fewer wave steps for the same pairs.  Per chunk every wave tests its 64 candidates per batch against the four blocks
// (footprint_hits_2x2 under a perfect pinhole), compacts the survivors into four byte lists in LDS (ballot + mbcnt), pads the shorter
// lists with the index of a NULL record (alpha = 0: never taken) and steps through them with one uniform counter: no per-lane queue
// state, one extra ds_read_u8 and a shift per step instead of the scalar bit walk.
static_assert(FCH == 128, "raster_fwd_quad_kernel: four byte lists of FCH entries = one ds_write_b64 per lane");
template <int KIND>
__global__ __launch_bounds__(RB, GSX_FWD_WAVES) void raster_fwd_quad_kernel(RasterArgs a, float* __restrict__ render_colors,
                                                                            float* __restrict__ render_alphas, int32_t* __restrict__ last_ids) {
    // AoS records, double buffered; [FCH] = the null record.  Pitch FIVE float4 (80 B): [0..3] the record, [4] = (rad2, k2, -, -) of the
    // footprint test.  A ds_read_b128 is served in groups of 16 lanes, and in the step loop a group holds lanes of TWO DPP rows reading
    // two different records: with a 64 B pitch their 4-bank windows coincide whenever the two list indices agree mod 4 (one extra LDS
    // cycle in a quarter of the group accesses: SQ_LDS_BANK_CONFLICT = 33 % of the kernel's LDS cycles in round 3); with 80 B they
    // coincide only for indices 16 apart.  The per-lane reads of the binning (stride = pitch) are conflict-free for the same reason
    // (64 B: four-way), and the cull plane of round 3 lives in the pad.
    __shared__ float4 s_rec[2][FCH + 1][5];
    __shared__ float s_bounds[4][4];
    __shared__ int s_wdone[2][4];
    GSX_BLOCK_CLOCK(1);
    // [wave][block][FCH]: indices into the chunk's records, consumed by the wave that wrote them (4 x FCH = 512 B per wave: one ds_write_b64 per
    // lane pre-fills them); + the byte the last list's look-ahead reads past its end
    __shared__ __attribute__((aligned(8))) uint8_t s_list_flat[4 * 4 * FCH + 8];
    uint8_t (*s_list)[4][FCH] = reinterpret_cast<uint8_t (*)[4][FCH]>(s_list_flat);
    const uint32_t cid = blockIdx.y;
    uint32_t tile_id;
    if (!swizzled_tile(blockIdx.x, a.tw * a.th, tile_id)) return;
    const uint32_t tile_y = tile_id / a.tw, tile_x = tile_id - tile_y * a.tw;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, qd = lane >> 4;
    // lane -> pixel: DPP row qd = 4x4 block (qd & 1, qd >> 1) of the wave's quadrant, 16 lanes = its pixels row-major
    const uint32_t j = tile_x * TILE + (wave & 1u) * 8u + (qd & 1u) * 4u + (lane & 3u);
    const uint32_t i = tile_y * TILE + (wave >> 1) * 8u + (qd >> 1) * 4u + ((lane >> 2) & 3u);
    const bool inside = i < a.H && j < a.W;
    const size_t pix = (size_t)cid * a.H * a.W + (size_t)i * a.W + j;
    const float* bg = a.backgrounds ? a.backgrounds + cid * 3 : nullptr;
    if (a.masks != nullptr && !a.masks[(size_t)cid * a.th * a.tw + tile_id]) {  // Fwd.cu:143-150
        if (inside)
            for (int k = 0; k < 3; ++k) render_colors[pix * 3 + k] = bg ? bg[k] : 0.f;
        return;
    }
    if (KIND == CAM_OPENCV_FISHEYE && a.tile_flags != nullptr && a.tile_flags[(size_t)cid * a.th * a.tw + tile_id]) return;  // generic kernel's tile
    const Camera<KIND> cam(a.cams, cid, a.W, a.H);
    float u, v, w;
    const bool ray_ok = pixel_ray(cam, i, j, u, v, w);
    bool done = !inside || !ray_ok;
    // (u, v) bounds of the valid pixels: per block (DPP row), per wave, per tile
    float qb[4];
    bool wide = false;
    {
        float bu = u, bv = v;
        if (KIND == CAM_OPENCV_FISHEYE) {
            wide = !done && w < 0.05f;
            const float iw = 1.f / fmaxf(w, 0.05f);
            bu = u * iw; bv = v * iw;
        }
        qb[0] = wide ? -INFINITY : (!done ? bu : INFINITY); qb[1] = wide ? INFINITY : (!done ? bu : -INFINITY);
        qb[2] = wide ? -INFINITY : (!done ? bv : INFINITY); qb[3] = wide ? INFINITY : (!done ? bv : -INFINITY);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            qb[0] = fminf(qb[0], __shfl_xor(qb[0], o)); qb[1] = fmaxf(qb[1], __shfl_xor(qb[1], o));
            qb[2] = fminf(qb[2], __shfl_xor(qb[2], o)); qb[3] = fmaxf(qb[3], __shfl_xor(qb[3], o));
        }
    }
    float bq[4][4];   // the four blocks' bounds, wave-uniform
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < 4; ++k) bq[q][k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qb[k]), 16 * q));
    float tb[4];
    {
        const float umin = fminf(fminf(bq[0][0], bq[1][0]), fminf(bq[2][0], bq[3][0])), umax = fmaxf(fmaxf(bq[0][1], bq[1][1]), fmaxf(bq[2][1], bq[3][1]));
        const float vmin = fminf(fminf(bq[0][2], bq[1][2]), fminf(bq[2][2], bq[3][2])), vmax = fmaxf(fmaxf(bq[0][3], bq[1][3]), fmaxf(bq[2][3], bq[3][3]));
        if (lane == 0) { s_bounds[wave][0] = umin; s_bounds[wave][1] = umax; s_bounds[wave][2] = vmin; s_bounds[wave][3] = vmax; }
        if (tid < 8) {   // the null records: lo' = -inf, unit denominator
            float4* nr = s_rec[tid >> 2][FCH];
            nr[tid & 3] = (tid & 3) == 1 ? make_float4(0.f, -INFINITY, 0.f, 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
        tb[0] = fminf(fminf(s_bounds[0][0], s_bounds[1][0]), fminf(s_bounds[2][0], s_bounds[3][0]));
        tb[1] = fmaxf(fmaxf(s_bounds[0][1], s_bounds[1][1]), fmaxf(s_bounds[2][1], s_bounds[3][1]));
        tb[2] = fminf(fminf(s_bounds[0][2], s_bounds[1][2]), fminf(s_bounds[2][2], s_bounds[3][2]));
        tb[3] = fmaxf(fmaxf(s_bounds[0][3], s_bounds[1][3]), fmaxf(s_bounds[2][3], s_bounds[3][3]));
    }
    // perfect pinhole: the blocks' rectangles are products of two u-ranges and two v-ranges (columns / rows of the quadrant)
    float xr[2][2], yr[2][2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        xr[k][0] = fminf(bq[k][0], bq[k + 2][0]); xr[k][1] = fmaxf(bq[k][1], bq[k + 2][1]);
        yr[k][0] = fminf(bq[2 * k][2], bq[2 * k + 1][2]); yr[k][1] = fmaxf(bq[2 * k][3], bq[2 * k + 1][3]);
    }
    const float ww = w * w;
    const bool no_cull = KIND == CAM_OPENCV_FISHEYE && !(tb[0] > -INFINITY);

    int32_t range_start, range_end;
    tile_list_range(a, cid, tile_x, tile_y, range_start, range_end);
    const int32_t n_chunks = (range_end - range_start + FCH - 1) / FCH;

    constexpr float K999 = 0.999f, LOG2_K999 = -0.0014434168696687174f, THR = (1.f / 255.f) / 0.999f;   // see raster_fwd_fast_kernel
    float T = 1.f;
    uint32_t cur_idx = 0;
    float out_r = 0.f, out_g = 0.f, out_b = 0.f;
    float thr = done ? INFINITY : THR;
    bool wave_done = __builtin_amdgcn_ballot_w64(!done) == 0ull;
    uint8_t* my_list = s_list[wave][qd];
    int32_t g_pre = 0;
    bool have = (int32_t)tid < FCH && range_start + (int32_t)tid < range_end;
    if (have) g_pre = a.flatten_ids[range_start + (int32_t)tid];
#ifdef GSX_STATS
    uint32_t st_tot[4] = {0u, 0u, 0u, 0u};
#endif
    for (int32_t b = 0; b < n_chunks; ++b) {
        const int buf = b & 1;
        const int32_t chunk_start = range_start + FCH * b;
        if (have) {
            StagedRec sr;
            stage_one(a, tb, g_pre, sr, tile_x, tile_y);
            if (no_cull) sr.cull.z = INFINITY;
            sr.r1.y -= LOG2_K999; sr.r2.w *= K999; sr.r3.x *= K999; sr.r3.y *= K999;
            s_rec[buf][tid][0] = sr.r0; s_rec[buf][tid][1] = sr.r1; s_rec[buf][tid][2] = sr.r2; s_rec[buf][tid][3] = sr.r3;
            s_rec[buf][tid][4] = make_float4(sr.cull.z, sr.cull.w, 0.f, 0.f);
        }
        if (lane == 0) s_wdone[buf][wave] = wave_done ? 1 : 0;
        __syncthreads();
        if (s_wdone[buf][0] & s_wdone[buf][1] & s_wdone[buf][2] & s_wdone[buf][3]) break;  // Fwd.cu:188-190
        have = (b + 1 < n_chunks) && (int32_t)tid < FCH && (chunk_start + FCH + (int32_t)tid < range_end);
        if (have) g_pre = a.flatten_ids[chunk_start + FCH + (int32_t)tid];  // in flight during the pixel loop
        if (wave_done) continue;
        const int32_t chunk_size = min(FCH, range_end - chunk_start);
        // the four lists of this wave for the chunk: every candidate (one per lane, 64 per batch) against the four blocks, survivors
        // compacted per block in list order; the lists were pre-filled with the null record's index, so the shorter ones idle to the end
        reinterpret_cast<unsigned long long*>(&s_list[wave][0][0])[lane] = 0x0101010101010101ull * (unsigned long long)FCH;
        uint32_t cnt[4] = {0u, 0u, 0u, 0u};
        for (int32_t sub = 0; sub < chunk_size; sub += 64) {
            uint32_t hits = 0u;
            if (sub + (int32_t)lane < chunk_size) {
                const float4 q0 = s_rec[buf][sub + lane][0], q1 = s_rec[buf][sub + lane][1], q4 = s_rec[buf][sub + lane][4];   // (u0, v0, l00, l01), (l11, ..), (rad2, k2)
                const float4 c = make_float4(q0.x, q0.y, q4.x, q4.y);
                if (KIND == CAM_PERFECT_PINHOLE) {
                    hits = footprint_hits_2x2(c, q0.z, q0.w, q1.x, xr, yr);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) hits |= footprint_hits(c, q0.z, q0.w, q1.x, bq[q][0], bq[q][1], bq[q][2], bq[q][3]) ? (1u << q) : 0u;
                }
            }
            GSX_STAT_ADD(1, min(64, chunk_size - sub));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool h = (hits >> q) & 1u;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(h);
                const uint32_t pos = cnt[q] + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (h) s_list[wave][q][pos] = (uint8_t)(sub + (int32_t)lane);
                cnt[q] += (uint32_t)__popcll(m);
            }
        }
        const uint32_t steps = max(max(cnt[0], cnt[1]), max(cnt[2], cnt[3]));
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the lists are private to the wave
        GSX_STAT_ADD(0, steps);
        GSX_STAT_ADD(4, cnt[0] + cnt[1] + cnt[2] + cnt[3]);      // sum of the four lists' lengths: / (4 x steps) = how full the four rows run
#ifdef GSX_STATS
        st_tot[0] += cnt[0]; st_tot[1] += cnt[1]; st_tot[2] += cnt[2]; st_tot[3] += cnt[3];
#endif
        uint32_t cur = 0xFFFFFFFFu;   // record index (inside the chunk) of the last Gaussian this pixel took
        uint32_t t_next = my_list[0];
        for (uint32_t k = 0; k < steps; ++k) {
            // one compositing step: every lane evaluates the Gaussian of ITS block's list for its pixel (same instruction sequence as
            // raster_fwd_fast_kernel's).  (Round 5, measured and not kept: the record address 80 t + base as two v_lshl_add_u32 instead of hipcc's
            // v_and + v_mad_u32_u24 — the pattern probe prices an isolated v_mad_u32_u24 at 4 plain instructions — A/B 0.1970 -> 0.1978 ms:
            // the step is co-limited by its 15 LDS cycles per wave, DESIGN.md §4 / NOTES.md.)
            const uint32_t t = t_next;
            t_next = my_list[k + 1];                     // (one past the end at the last step: inside the LDS arrays, never used)
            const float4* rp = s_rec[buf][t];
            const float4 r0 = rp[0], r1 = rp[1], r2 = rp[2], r3 = rp[3];
            const float du = KIND == CAM_OPENCV_FISHEYE ? fmaf(-w, r0.x, u) : u - r0.x;   // fisheye: unnormalised ray (u, v, w), see fast_alpha_ray
            const float dv = KIND == CAM_OPENCV_FISHEYE ? fmaf(-w, r0.y, v) : v - r0.y;
            const float t0 = fmaf(r0.w, dv, r0.z * du);
            const float t1 = r1.x * dv;
            const float num2 = fmaf(t0, t0, t1 * t1);
            const float den = KIND == CAM_OPENCV_FISHEYE
                                  ? fmaf(du, fmaf(r2.x, du, fmaf(r2.y, dv, r1.z * w)), fmaf(dv, fmaf(r2.z, dv, r1.w * w), ww))
                                  : fmaf(du, fmaf(r2.x, du, fmaf(r2.y, dv, r1.z)), fmaf(dv, fmaf(r2.z, dv, r1.w), 1.f));
            const float ap = __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(fmaf(-num2, __builtin_amdgcn_rcpf(den), r1.y)), 0.f, 1.f);
            bool take = ap >= thr;                       // alpha >= 1/255 and the pixel is not finished (Fwd.cu:240)
            float wgt = take ? ap * T : 0.f;             // alpha T / 0.999
            T = fmaf(-K999, wgt, T);                     // T (1 - alpha), in place
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(T <= 1e-4f) != 0ull, 0)) {
                const bool stop = T <= 1e-4f;            // this pixel does NOT take the Gaussian (Fwd.cu:245-248): see raster_fwd_fast_kernel
                take = take && !stop;
                T = stop ? fmaf(K999, wgt, T) : T;
                wgt = stop ? 0.f : wgt;
                thr = stop ? INFINITY : thr;
            }
            out_r = fmaf(r2.w, wgt, out_r); out_g = fmaf(r3.x, wgt, out_g); out_b = fmaf(r3.y, wgt, out_b);
            cur = take ? t : cur;
#ifdef GSX_STATS
            { const unsigned long long cc = __builtin_amdgcn_ballot_w64(take); GSX_STAT_ADD(2, cc != 0ull); GSX_STAT_ADD(3, __popcll(cc)); }
#endif
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the next chunk's pre-fill stays behind this chunk's reads
        cur_idx = cur != 0xFFFFFFFFu ? (uint32_t)chunk_start + cur : cur_idx;
        if (__builtin_amdgcn_ballot_w64(thr < INFINITY) == 0ull) wave_done = true;   // all 64 pixels finished
    }
    GSX_STAT_ADD(5, max(max(st_tot[0], st_tot[1]), max(st_tot[2], st_tot[3])));   // steps if the lists ran on across chunk boundaries (per wave and tile: the longest list)
    if (inside) {
        render_alphas[pix] = 1.f - T;
        render_colors[pix * 3] = bg ? out_r + T * bg[0] : out_r;
        render_colors[pix * 3 + 1] = bg ? out_g + T * bg[1] : out_g;
        render_colors[pix * 3 + 2] = bg ? out_b + T * bg[2] : out_b;
        last_ids[pix] = (int32_t)cur_idx;
    }
}

// ---- forward, two pixels per lane and eight lists per wave ("pair" variant, round 5; perfect pinhole) -----------------------------
// The four-list kernel reads one 56 B record from LDS per (lane, step) and evaluates ONE pixel with it: 15 LDS-array cycles per step per
// wave, four SIMDs on one LDS — the LDS is 73 % busy beside a VALU that is 71 % busy, and every lane runs one dependent chain (list entry ->
// record -> rcp -> exp2 -> compare -> T) per step (profiles/r05_valu_issue.md).  Here a lane owns TWO horizontally adjacent pixels: the
// tile is two waves (wave w = its rows 8 w .. 8 w + 7), eight lanes hold a 4x4 block (lane k of the eight: row k >> 1, columns 2 (k & 1),
// 2 (k & 1) + 1), a wave walks the lists of its EIGHT blocks (4 across, 2 down) with one uniform step counter.  Per step a lane reads its
// block's record once and composites two pixels: half the LDS reads per pair, two independent chains per lane, and the terms of the pair
// that depend on dv alone (a lane's pixels share their image row) are formed once — by the compiler's CSE on the SAME expressions as the
// other two kernels: every (pixel, Gaussian) pair sees the same instructions in the same order and the outputs are bit-identical
// (tests/test_gpu_fused.py::test_forward_kernels_are_bit_identical).  46 VALU per step of two pixels (four-list kernel: 27 per pixel).
// Chunks of PCH = 64 records (one binning batch), double buffered: 12.5 KB of LDS and 76 VGPRs per two-wave workgroup, 6 waves per SIMD (a
// chunk of 128 fills the eight lists better — 84 % instead of 79 % — and loses more to its 3.5 waves per SIMD: 0.234 against 0.219 ms).
// List entries are the record's BYTE OFFSET inside the chunk's buffer (16 bits): one add instead of a mask, a move and a v_mad_u32_u24.  (Entries
// that are the LDS address itself, no add at all, measured 3 % SLOWER, and so did an s_nop in the add's place, with or without aligned loops; a
// two-step software pipeline — the next record in flight while this one composites — 90 VGPRs, 52 VALU per step: +3 %.  NOTES.md N1.)
// Counters at S-1M (profiles/r05f_pmc_counters.md): 1.889 M steps of two pixels (four lists: 3.36 M of one), SQ_INSTS_VALU 124.6 M (127.8), LDS
// cycles 51.9 M (73.7), SALU 14.1 M (31) — the step loop is 87 M of the VALU instructions in either kernel, the other 37 M are binning (every
// candidate against 16 blocks: 22 M), staging and set-up.
// Measured, same box (NOTES.md N1): the op alone in a loop (tools/fwd_quad_ab.py, incl. record packing) S-1M 0.2197 (four lists) -> 0.2094 ms,
// S-5M @4K 0.875 (one list) / 0.93 (four lists) -> 0.834, a saturated 1080p frame 0.2488 (one list) -> 0.2390, large footprints on 32-pixel
// lists 0.1272 (one list) -> 0.1280, the saturated frame at 640 x 360 0.0555 (one list) -> 0.0632 (a wave stops when all of its 128 pixels
// are finished, and 920 tiles of two waves do not fill the chip).  INSIDE the training step, where the chip sits at its power limit all the
// time, S-1M's kernel goes 198.9 -> 197.0 us only (tools/ktrace_fwd_modes.sh) — and the loop's gain disappears the same way when a 2 GiB
// streaming kernel runs in front of every timed launch: under sustained load a kernel is paid in energy, not in utilisation.  S-5M's forward
// inside its step: 0.842 -> 0.792 ms.
constexpr int PB = 128;    // threads per workgroup of the pair kernel
constexpr int PCH = 64;    // records per chunk
// footprint_hits for the NX x NY blocks whose rectangles are products of NX u-ranges and NY v-ranges (hit[sy * NX + sx]): the 1-D pieces once
// per range; the results stay compare results (the caller ballots them: no packing into bits and back)
template <int NX, int NY>
GSX_DEV void footprint_hits_grid(float4 c, float l00, float l01, float l11, const float (&xr)[NX][2], const float (&yr)[NY][2], bool (&hit)[NX * NY]) {
    float xa[NX], xb[NX], lxa[NX], lxb[NX], lxc[NX], kxc[NX], ya[NY], yb[NY], m[NY], e1s[NY];
#pragma unroll
    for (int k = 0; k < NX; ++k) {
        xa[k] = xr[k][0] - c.x; xb[k] = xr[k][1] - c.x;
        const float xc = __builtin_amdgcn_fmed3f(0.f, xa[k], xb[k]);
        lxa[k] = l00 * xa[k]; lxb[k] = l00 * xb[k]; lxc[k] = l00 * xc; kxc[k] = c.w * xc;
    }
#pragma unroll
    for (int k = 0; k < NY; ++k) {
        ya[k] = yr[k][0] - c.y; yb[k] = yr[k][1] - c.y;
        const float yc = __builtin_amdgcn_fmed3f(0.f, ya[k], yb[k]);
        m[k] = l01 * yc;
        const float e1 = l11 * yc;
        e1s[k] = e1 * e1;
    }
#pragma unroll
    for (int sy = 0; sy < NY; ++sy)
#pragma unroll
        for (int sx = 0; sx < NX; ++sx) {
            const float t = __builtin_amdgcn_fmed3f(-m[sy], lxa[sx], lxb[sx]) + m[sy];
            const float n1 = fmaf(t, t, e1s[sy]);
            const float ys = __builtin_amdgcn_fmed3f(kxc[sx], ya[sy], yb[sy]);
            const float t2 = fmaf(l01, ys, lxc[sx]), e2 = l11 * ys;
            const float n2 = fmaf(t2, t2, e2 * e2);
            hit[sy * NX + sx] = !(fminf(n1, n2) > c.z);
        }
}

#ifndef GSX_PAIR_WAVES
#define GSX_PAIR_WAVES 6   // (7 / 8: 0.2277 / 0.2201 against 0.2132 ms; 5 / 4: +2 %)
#endif
__global__ __launch_bounds__(PB, GSX_PAIR_WAVES) void raster_fwd_pair_kernel(RasterArgs a, float* __restrict__ render_colors,
                                                                             float* __restrict__ render_alphas, int32_t* __restrict__ last_ids) {
    constexpr int KIND = CAM_PERFECT_PINHOLE;
    constexpr uint32_t PITCH = 80u;
    static_assert((PCH + 1) * PITCH < 65536u, "list entries are 16-bit byte offsets");
    __shared__ float4 s_rec[2][PCH + 1][5];   // as in raster_fwd_quad_kernel: 80 B pitch, [PCH] = the null record, [.][4] = (rad2, k2, -, -)
    __shared__ float s_bounds[2][2];
    __shared__ int s_wdone[2][2];
    // [wave][block][PCH + 8]: a list pitch of 144 B, not 128 — the step loop's ds_read_u16 is serviced in two groups of 32 lanes = four blocks each, banks
    // (a / 4) mod 32: at a pitch of 128 B the eight lists' k-th entries sat on ONE bank (a 4-way conflict on every step: 6 of the kernel's 7.9 conflict
    // cycles per step, profiles/r05f_pmc_counters.md); 144 B puts them four banks apart.  Entry [PCH] of a list is what the last step's look-ahead reads.
    constexpr int LP = PCH + 8;
    __shared__ __attribute__((aligned(16))) uint16_t s_list_flat[2 * 8 * LP];
    uint16_t (*s_list)[8][LP] = reinterpret_cast<uint16_t (*)[8][LP]>(s_list_flat);
    const uint32_t cid = blockIdx.y;
    uint32_t tile_id;
    if (!swizzled_tile(blockIdx.x, a.tw * a.th, tile_id)) return;
    const uint32_t tile_y = tile_id / a.tw, tile_x = tile_id - tile_y * a.tw;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, q8 = lane >> 3, k8 = lane & 7u;
    const uint32_t uwave = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);
    // lane -> its two pixels: block q8 = (column q8 & 3, row q8 >> 2) of the wave's 16 x 8 half tile, lane k8 of the block = row k8 >> 1, columns 2 (k8 & 1) + {0, 1}
    const uint32_t j0 = tile_x * TILE + (q8 & 3u) * 4u + (k8 & 1u) * 2u;
    const uint32_t i = tile_y * TILE + wave * 8u + (q8 >> 2) * 4u + (k8 >> 1);
    const bool inside[2] = {i < a.H && j0 < a.W, i < a.H && j0 + 1u < a.W};
    const size_t pix0 = (size_t)cid * a.H * a.W + (size_t)i * a.W + j0;
    const float* bg = a.backgrounds ? a.backgrounds + cid * 3 : nullptr;
    if (a.masks != nullptr && !a.masks[(size_t)cid * a.th * a.tw + tile_id]) {  // Fwd.cu:143-150
#pragma unroll
        for (int p = 0; p < 2; ++p)
            if (inside[p])
                for (int k = 0; k < 3; ++k) render_colors[(pix0 + p) * 3 + k] = bg ? bg[k] : 0.f;
        return;
    }
    const Camera<KIND> cam(a.cams, cid, a.W, a.H);
    float u[2], v;
    pixel_uv(cam, i, j0, u[0], v);
    { float v1; pixel_uv(cam, i, j0 + 1u, u[1], v1); }   // (same row: v1 == v)
    // The blocks' rectangles of pixel centres are products of four u-ranges (block columns) and two v-ranges (block rows), clipped to the image:
    // u depends on the column only, v on the row only, so the ends of a range are the (u, v) of the lanes that own its first and last valid
    // pixel — wave-uniform lane numbers, read with v_readlane (no reductions over the lanes); a range without a valid pixel is empty (inf, -inf)
    const int32_t wx0 = (int32_t)(tile_x * TILE), wy0 = (int32_t)(tile_y * TILE + uwave * 8u);
    float xr[4][2], yr[2][2];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int32_t last = min(3, (int32_t)a.W - 1 - (wx0 + 4 * k));   // last valid column of block column k (< 0: none)
        const int32_t ll = max(last, 0);
        const float first_u = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, u[0]), 8 * k));
        const float lu0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, u[0]), 8 * k + (ll >> 1)));
        const float lu1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, u[1]), 8 * k + (ll >> 1)));
        xr[k][0] = last >= 0 ? first_u : INFINITY;
        xr[k][1] = last >= 0 ? ((ll & 1) ? lu1 : lu0) : -INFINITY;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int32_t last = min(3, (int32_t)a.H - 1 - (wy0 + 4 * k));   // last valid row of block row k
        const int32_t ll = max(last, 0);
        const float first_v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32 * k));
        const float last_v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32 * k + 2 * ll));
        yr[k][0] = last >= 0 ? first_v : INFINITY;
        yr[k][1] = last >= 0 ? last_v : -INFINITY;
    }
    float tb[4];   // the tile's bounds (footprint()): its columns are this wave's, its rows both waves'
    {
        tb[0] = xr[0][0];
        tb[1] = fmaxf(fmaxf(xr[0][1], xr[1][1]), fmaxf(xr[2][1], xr[3][1]));
        if (lane == 0) { s_bounds[wave][0] = fminf(yr[0][0], yr[1][0]); s_bounds[wave][1] = fmaxf(yr[0][1], yr[1][1]); }
        if (tid < 10) {   // the null records: lo' = -inf, unit denominator, empty footprint (rad2 = -1: what the binning's idle lanes test)
            const uint32_t part = tid % 5u;
            s_rec[tid / 5u][PCH][part] = part == 1u ? make_float4(0.f, -INFINITY, 0.f, 0.f) : (part == 4u ? make_float4(-1.f, 0.f, 0.f, 0.f) : make_float4(0.f, 0.f, 0.f, 0.f));
        }
        __syncthreads();
        tb[2] = fminf(s_bounds[0][0], s_bounds[1][0]); tb[3] = fmaxf(s_bounds[0][1], s_bounds[1][1]);
    }

    int32_t range_start, range_end;
    tile_list_range(a, cid, tile_x, tile_y, range_start, range_end);
    const int32_t n_chunks = (range_end - range_start + PCH - 1) / PCH;

    constexpr float K999 = 0.999f, LOG2_K999 = -0.0014434168696687174f, THR = (1.f / 255.f) / 0.999f;   // see raster_fwd_fast_kernel
    float T[2] = {1.f, 1.f};
    uint32_t cur_idx[2] = {0u, 0u};
    float out_r[2] = {0.f, 0.f}, out_g[2] = {0.f, 0.f}, out_b[2] = {0.f, 0.f};
    float thr[2] = {inside[0] ? THR : INFINITY, inside[1] ? THR : INFINITY};
    bool wave_done = __builtin_amdgcn_ballot_w64(inside[0] || inside[1]) == 0ull;
    const uint16_t* my_list = s_list[wave][q8];
#ifdef GSX_STATS
    uint32_t st_tot[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
#endif
    // wave 0 stages the chunks, one record per lane (the waves taking turns measured 1 % slower)
    const bool stager = wave == 0u;
    int32_t g_pre = 0;
    bool have = stager && range_start + (int32_t)lane < range_end;
    if (have) g_pre = a.flatten_ids[range_start + (int32_t)lane];
    for (int32_t b = 0; b < n_chunks; ++b) {
        const int buf = b & 1;
        const int32_t chunk_start = range_start + PCH * b;
        if (have) {
            StagedRec sr;
            stage_one(a, tb, g_pre, sr, tile_x, tile_y);
            sr.r1.y -= LOG2_K999; sr.r2.w *= K999; sr.r3.x *= K999; sr.r3.y *= K999;
            s_rec[buf][lane][0] = sr.r0; s_rec[buf][lane][1] = sr.r1; s_rec[buf][lane][2] = sr.r2; s_rec[buf][lane][3] = sr.r3;
            s_rec[buf][lane][4] = make_float4(sr.cull.z, sr.cull.w, 0.f, 0.f);
        }
        if (lane == 0) s_wdone[buf][wave] = wave_done ? 1 : 0;
        __syncthreads();
        if (s_wdone[buf][0] & s_wdone[buf][1]) break;  // Fwd.cu:188-190
        have = stager && (b + 1 < n_chunks) && (chunk_start + PCH + (int32_t)lane < range_end);
        if (have) g_pre = a.flatten_ids[chunk_start + PCH + (int32_t)lane];  // in flight during the pixel loop
        if (wave_done) continue;
        const int32_t chunk_size = min(PCH, range_end - chunk_start);
        // the eight lists of this wave for the chunk, pre-filled with the null record's entry (the shorter ones idle to the end): 8 x 64 x 2 B = one ds_write_b128 per lane
        {
            const uint32_t fill = 0x00010001u * (PCH * PITCH);
            reinterpret_cast<uint4*>(&s_list[wave][q8][0])[k8] = make_uint4(fill, fill, fill, fill);   // lane (q8, k8): 16 B of list q8
        }
        // every lane tests a record — lanes behind the chunk's end the null record (empty footprint) —: no branch, the eight compare results
        // ARE the ballots and the counters stay wave-uniform
        const uint32_t cand = (int32_t)lane < chunk_size ? lane : (uint32_t)PCH;
        bool hit[8];
        {
            const float4 q0 = s_rec[buf][cand][0], q1 = s_rec[buf][cand][1], q4 = s_rec[buf][cand][4];   // (u0, v0, l00, l01), (l11, ..), (rad2, k2)
            footprint_hits_grid<4, 2>(make_float4(q0.x, q0.y, q4.x, q4.y), q0.z, q0.w, q1.x, xr, yr, hit);
        }
        // Round 6 (VERDICT r05 next #4b), built, measured, NOT the default (-DGSX_PAIR_ALIVE_MASK): a 4 x 4 block whose 16 pixels are all finished could
        // stop taking list entries (its lanes step through them with the threshold at +inf until the slowest pixel of the tile finishes).  Counted with
        // -DGSX_STATS (tools/ab_alive.sh, profiles/r06_processed_intersections.md): the mask saves 0.05 % of the steps at S-1M and 1.3 % at S-5M @4K
        // (92 % of its pixels saturate, but a tile's blocks finish within a chunk or two of each other and the tile exit already ends the walk after
        // 33 % of the list) — and costs the S-1M forward 3 % (0.2495 / 0.2535 -> 0.2571 / 0.2621 ms alternating libraries), S-5M -0.5 %.
#ifdef GSX_PAIR_ALIVE_MASK
        const unsigned long long alive = __builtin_amdgcn_ballot_w64(thr[0] < INFINITY || thr[1] < INFINITY);
#else
        const unsigned long long alive = ~0ull;
#endif
        uint32_t cnt[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const bool block_alive = ((alive >> (8 * q)) & 0xFFull) != 0ull;   // wave-uniform
            const unsigned long long m = block_alive ? __builtin_amdgcn_ballot_w64(hit[q]) : 0ull;
            const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (hit[q] && block_alive) s_list[wave][q][pos] = (uint16_t)(cand * PITCH);
            cnt[q] = (uint32_t)__popcll(m);
        }
        const uint32_t steps = (uint32_t)__builtin_amdgcn_readfirstlane((int)max(max(max(cnt[0], cnt[1]), max(cnt[2], cnt[3])), max(max(cnt[4], cnt[5]), max(cnt[6], cnt[7]))));   // (wave-uniform: the step counter stays on the SALU)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the lists are private to the wave
        GSX_STAT_ADD(0, steps);
        GSX_STAT_ADD(1, chunk_size);
        GSX_STAT_ADD(4, cnt[0] + cnt[1] + cnt[2] + cnt[3] + cnt[4] + cnt[5] + cnt[6] + cnt[7]);   // / (8 x steps) = how full the eight lists run
#ifdef GSX_STATS
#pragma unroll
        for (int q = 0; q < 8; ++q) st_tot[q] += cnt[q];
#endif
        uint32_t cur[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};   // list entry (the record's byte offset inside the chunk's buffer) of the last Gaussian each pixel took
        uint32_t t_next = my_list[0];
        for (uint32_t k = 0; k < steps; ++k) {
            const uint32_t t = t_next;
            t_next = my_list[k + 1];                     // (one past the end at the last step: inside the LDS arrays, never used)
            const float4* rp = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(&s_rec[buf][0][0]) + t);
            const float4 r0 = rp[0], r1 = rp[1], r2 = rp[2], r3 = rp[3];
            bool take[2];
            float wgt[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {   // the same expressions as raster_fwd_quad_kernel's step, per pixel
                const float du = u[p] - r0.x;
                const float dv = v - r0.y;
                const float t0 = fmaf(r0.w, dv, r0.z * du);
                const float t1 = r1.x * dv;
                const float num2 = fmaf(t0, t0, t1 * t1);
                const float den = fmaf(du, fmaf(r2.x, du, fmaf(r2.y, dv, r1.z)), fmaf(dv, fmaf(r2.z, dv, r1.w), 1.f));
                const float ap = __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(fmaf(-num2, __builtin_amdgcn_rcpf(den), r1.y)), 0.f, 1.f);
                take[p] = ap >= thr[p];                  // alpha >= 1/255 and the pixel is not finished (Fwd.cu:240)
                wgt[p] = take[p] ? ap * T[p] : 0.f;      // alpha T / 0.999
                T[p] = fmaf(-K999, wgt[p], T[p]);        // T (1 - alpha), in place
            }
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(T[0] <= 1e-4f || T[1] <= 1e-4f) != 0ull, 0)) {
#ifdef GSX_STATS
                {   // how often a wave takes the stop branch, and for how many pixels (the ballots outside the macro's `lane == 0` region)
                    const uint32_t n_stop = (uint32_t)(__popcll(__builtin_amdgcn_ballot_w64(T[0] <= 1e-4f)) + __popcll(__builtin_amdgcn_ballot_w64(T[1] <= 1e-4f)));
                    GSX_STAT_ADD(6, 1); GSX_STAT_ADD(7, n_stop);
                }
#endif
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const bool stop = T[p] <= 1e-4f;     // this pixel does NOT take the Gaussian (Fwd.cu:245-248): see raster_fwd_fast_kernel
                    take[p] = take[p] && !stop;
                    T[p] = stop ? fmaf(K999, wgt[p], T[p]) : T[p];
                    wgt[p] = stop ? 0.f : wgt[p];
                    thr[p] = stop ? INFINITY : thr[p];
                }
            }
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                out_r[p] = fmaf(r2.w, wgt[p], out_r[p]); out_g[p] = fmaf(r3.x, wgt[p], out_g[p]); out_b[p] = fmaf(r3.y, wgt[p], out_b[p]);
                cur[p] = take[p] ? t : cur[p];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the next chunk's pre-fill stays behind this chunk's reads
#pragma unroll
        for (int p = 0; p < 2; ++p) cur_idx[p] = cur[p] != 0xFFFFFFFFu ? (uint32_t)chunk_start + cur[p] / PITCH : cur_idx[p];
        if (__builtin_amdgcn_ballot_w64(thr[0] < INFINITY || thr[1] < INFINITY) == 0ull) wave_done = true;   // all 128 pixels finished
    }
    GSX_STAT_ADD(5, max(max(max(st_tot[0], st_tot[1]), max(st_tot[2], st_tot[3])), max(max(st_tot[4], st_tot[5]), max(st_tot[6], st_tot[7]))));   // steps if the lists ran on across chunk boundaries
#pragma unroll
    for (int p = 0; p < 2; ++p)
        if (inside[p]) {
            const size_t pix = pix0 + (size_t)p;
            render_alphas[pix] = 1.f - T[p];
            render_colors[pix * 3] = bg ? out_r[p] + T[p] * bg[0] : out_r[p];
            render_colors[pix * 3 + 1] = bg ? out_g[p] + T[p] * bg[1] : out_g[p];
            render_colors[pix * 3 + 2] = bg ? out_b[p] + T[p] * bg[2] : out_b[p];
            last_ids[pix] = (int32_t)cur_idx[p];
        }
}

// forward workspace, from its 256 B aligned base: packed records [C*N] x 64 B | fisheye: "no chart" bytes [C*N] | tile flags
// ... | chain heads [NSUB][C*N] int32 of the backward's per-(camera, Gaussian) record chains: set to -1 by whoever packs the records (the
// fused front end, pack_records_kernel) and put back to -1 by the gather kernel that walks the chains, so a backward on the forward's
// workspace needs no memset launch
// ... | 256 B: word [1] = the mode of the head planes (0 = chains, 1 = ranges: see gsx_raster_common.hpp) | ranges: first record slot of every wave of 64 Gaussians
static size_t head_planes_bytes(uint32_t C, uint32_t N) { return align256((size_t)C * N * 4 * NSUB) + 256 + align256((((size_t)C * N + 63) / 64 + 64) * 4); }   // the planes + the mode word + the waves' first slots (ranges)
size_t raster_fwd_fast_workspace_bytes(uint32_t C, uint32_t N) { return 256 + (size_t)C * N * 64 + align256((size_t)C * N) + FAST_FLAG_BYTES + head_planes_bytes(C, N); }
static uint8_t* ws_bad(const float4* packed, uint32_t C, uint32_t N) { return (uint8_t*)packed + (size_t)C * N * 64; }
static uint8_t* ws_flags(const float4* packed, uint32_t C, uint32_t N) { return ws_bad(packed, C, N) + align256((size_t)C * N); }
int32_t* raster_fwd_fast_heads(const float4* packed, uint32_t C, uint32_t N) { return (int32_t*)(ws_flags(packed, C, N) + FAST_FLAG_BYTES); }
uint32_t* raster_fwd_fast_alloc(const float4* packed, uint32_t C, uint32_t N) { return (uint32_t*)((char*)raster_fwd_fast_heads(packed, C, N) + align256((size_t)C * N * 4 * NSUB)); }   // [1] mode word; + 64 words: the waves' first slots

// mode of the head planes behind `ws_head` (REC_MODE_*: written by whoever packed the workspace); wave-uniform scalar load
GSX_DEV bool record_ranges(const RasterArgs& a, const int32_t* ws_head) {
    return reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(ws_head) + align256((size_t)a.C * a.N * 4 * NSUB))[1] == REC_MODE_RANGES;
}
// ranges: first slot of the run of (camera, Gaussian) g = first slot of its wave of 64 Gaussians + its offset inside the wave (plane 0)
GSX_DEV int32_t range_first_slot(const RasterArgs& a, const int32_t* ws_head, size_t g) {
    const uint32_t* wave_first = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(ws_head) + align256((size_t)a.C * a.N * 4 * NSUB) + 256);
    return (int32_t)wave_first[g >> 6] + ws_head[g];
}
// slot of the moment record of (tile, list entry `isect`) for (camera, Gaussian) g, claimed with ONE returning atomic either way:
// ranges — the next slot of the Gaussian's own run (no link); chains — the entry's tile-major slot, linked in front of the Gaussian's chain
GSX_DEV int32_t claim_record_slot(const RasterArgs& a, int32_t* ws_head, bool ranges, int32_t g, int32_t isect, uint32_t tile_x, uint32_t tile_y, int32_t& link) {
    const size_t cn = (size_t)a.C * a.N;
    if (ranges) {
        link = -1;
        return range_first_slot(a, ws_head, (size_t)g) + atomicAdd(&ws_head[cn + (size_t)g], 1);   // plane 1 = records claimed so far
    }
    const int32_t slot = tile_record_slot(a, isect, tile_x, tile_y);
    link = atomicExch(&ws_head[(size_t)tile_chain(a, tile_x, tile_y) * cn + (size_t)g], slot);
    return slot;
}

// packs the records (and, for a fisheye, flags the tiles the fast kernels must leave to the reference-order kernels)
static const float4* pack_into(int kind, RasterArgs& a, void* base, hipStream_t st, int32_t* heads = nullptr) {
    float4* packed = (float4*)(((uintptr_t)base + 255) & ~(uintptr_t)255);
    const bool fisheye = kind == CAM_OPENCV_FISHEYE;
    hipLaunchKernelGGL(pack_records_kernel, dim3((a.N + 255u) / 256u, a.C), dim3(256), 0, st, a, packed, heads, fisheye ? ws_bad(packed, a.C, a.N) : nullptr);
    a.packed = packed;
    if (fisheye) {
        uint8_t* flags = ws_flags(packed, a.C, a.N);
        hipLaunchKernelGGL(tile_flag_kernel, dim3((a.tw * a.th + 3u) / 4u, a.C), dim3(256), 0, st, a, (const uint8_t*)ws_bad(packed, a.C, a.N), flags);
        a.tile_flags = flags;
    }
    return packed;
}

const uint8_t* launch_raster_fwd_fast(int kind, RasterArgs a, float* renders, float* alphas, int32_t* last_ids, void* workspace,
                                      size_t workspace_bytes, hipStream_t st, bool records_ready) {
    (void)workspace_bytes;
    const uint32_t n_tiles = a.tw * a.th;
    const dim3 grid(((n_tiles + 7u) / 8u) * 8u, a.C), block(RB);
    if (records_ready) a.packed = (const float4*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);   // written by the fused front end (records + list heads)
    else pack_into(kind, a, workspace, st, raster_fwd_fast_heads((const float4*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255), a.C, a.N));
    // Which forward kernel: four lists per wave pay where a Gaussian that reaches a quadrant misses part of its four blocks — small
    // footprints.  Measured (tools/fwd_quad_ab.py, same box): S-1M (3.4 tiles per Gaussian) 0.272 -> 0.232 ms, its fisheye twin 0.320 ->
    // 0.265; S-5M @4K (5.3 tiles) -5 .. +7 %, a saturated scene (5.9) +3 %, large footprints with 32-pixel lists +18 % (every block sees
    // nearly every survivor and the four block tests per candidate are pure cost).  Rule: 16-pixel lists and at most 4.5 intersections
    // per (camera, Gaussian) on average; GSX_FWD=quad|wave forces one (tests, A/B tools; read per launch).
    bool quad = a.lshift == 0u && 2 * a.n_isects_expected <= 9 * (int64_t)a.C * (int64_t)a.N;
    if (const char* e = test_switch("GSX_FWD")) quad = std::string(e) == "quad" ? true : (std::string(e) == "wave" ? false : quad);
#define GSX_BLEND_FWD(KERNEL)                                                                                                            \
    do {                                                                                                                                 \
        if (kind == CAM_PERFECT_PINHOLE) hipLaunchKernelGGL(HIP_KERNEL_NAME(KERNEL<CAM_PERFECT_PINHOLE>), grid, block, 0, st, a, renders, alphas, last_ids); \
        else if (kind == CAM_OPENCV_PINHOLE) hipLaunchKernelGGL(HIP_KERNEL_NAME(KERNEL<CAM_OPENCV_PINHOLE>), grid, block, 0, st, a, renders, alphas, last_ids); \
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(KERNEL<CAM_OPENCV_FISHEYE>), grid, block, 0, st, a, renders, alphas, last_ids);               \
    } while (0)
    // Two pixels per lane / eight lists per wave (raster_fwd_pair_kernel, perfect pinhole): frames with 16-pixel lists whose grid fills the chip with
    // two-wave workgroups (>= 2048 tiles) and up to 8 intersections per Gaussian.  Same box, op incl. record packing: S-1M 0.2197 (four lists) -> 0.2094,
    // S-5M @4K 0.875 (one list) -> 0.834, a saturated 1080p frame (6.1 intersections per Gaussian) 0.2488 (one list) -> 0.2390; NOT the same frame at
    // 640 x 360 (920 tiles: 0.0555 -> 0.0632) and not 32-pixel lists (0.1272 -> 0.1280).  GSX_FWD=pair forces it for any perfect-pinhole frame.
    bool pair = a.lshift == 0u && kind == CAM_PERFECT_PINHOLE && n_tiles >= 2048u && a.n_isects_expected <= 8 * (int64_t)a.C * (int64_t)a.N;
    if (const char* e = test_switch("GSX_FWD")) pair = std::string(e) == "pair" ? kind == CAM_PERFECT_PINHOLE : (std::string(e) == "quad" || std::string(e) == "wave" ? false : pair);
    if (pair) hipLaunchKernelGGL(raster_fwd_pair_kernel, grid, dim3(PB), 0, st, a, renders, alphas, last_ids);
    else if (quad) GSX_BLEND_FWD(raster_fwd_quad_kernel);
    else GSX_BLEND_FWD(raster_fwd_fast_kernel);
#undef GSX_BLEND_FWD
    return a.tile_flags;
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
// With D = N/Dn, N = |du B0 + dv B1|^2, Dn = |h + a0 du + a1 dv|^2 every parameter gradient is LINEAR in 11
// pixel moments of each (camera, Gaussian)
//     Ma = sum a {du^2, du dv, dv^2, du, dv},   Mb = sum b {1, du, dv, du^2, du dv, dv^2},
//     a = (dL/dD) / den',  b = a D,
// which the kernels take in the WHITENED offsets x0 = l00 du + l01 dv, x1 = l11 dv the alpha evaluation forms anyway (a thin footprint makes
// the (du, dv) moments nearly rank one and loses the short axes' gradients to fp32; gsx_record.hpp: moments_to_gradients),
// plus v_rgb[3] and the opacity term: 15 sums.  The coefficients of that linear map depend on the Gaussian and
// the camera only — not on the tile — so the kernel accumulates moments and the chain rule runs ONCE per
// (camera, Gaussian) afterwards (gsx_bwd_gather_kernel), not once per (tile, Gaussian).
//
// Per evaluated (wave, Gaussian) every lane produces its pixel's 15 terms; they are reduced over the wave by a
// multi-value butterfly (v_permlane32_swap / v_permlane16_swap halve the register count at each level: 40 VALU for
// 16 values instead of 96 for 16 independent 6-step reductions) and added to the chunk's LDS accumulator.
// (A two-phase variant — park per-pair weights in LDS, then fold pixels per Gaussian with lanes = Gaussians —
// was measured too: 26 % fewer VALU instructions but twice the LDS traffic, 1.35 ms vs 1.24 ms; not kept.)
// At the end of a chunk one thread per touched Gaussian writes its 64 B moment record to the workspace and chains
// it into the Gaussian's list with one returning exchange: no float atomics reach memory (on MI355X the 8 XCD L2s
// are not coherent, device-scope float atomics are served memory-side and 14 of them per (tile, Gaussian) cost
// more than all the arithmetic of the kernel).
constexpr int NMOM = 16;
#ifndef GSX_BCH
#define GSX_BCH 128
#endif
constexpr int BCH = GSX_BCH;  // Gaussians per backward chunk

#if defined(GSX_ABLATE)
#define GSX_GATOMIC(p, v) atomicAdd((p), (v))
#endif

template <int KIND>
__global__ __launch_bounds__(RB, GSX_BWD_WAVES) void raster_bwd_fast_kernel(RasterArgs a, const float* __restrict__ render_alphas,
                                                                            const int32_t* __restrict__ last_ids,
                                                                            const float* __restrict__ v_render_colors,
                                                                            const float* __restrict__ v_render_alphas,
                                                                            float4* __restrict__ ws_rec, int32_t* __restrict__ ws_head) {
    __shared__ float4 s_rec[BCH][4];
    __shared__ float4 s_cull[BCH];
    __shared__ float s_acc[NMOM][BCH];
    __shared__ int32_t s_gid[BCH];
    __shared__ unsigned long long s_touched[(BCH + 63) / 64];
    __shared__ float s_bounds[4][4];
    __shared__ int32_t s_blockmax;
    GSX_BLOCK_CLOCK(3);
    const uint32_t cid = blockIdx.y;
    const bool ranges = record_ranges(a, ws_head);   // how this workspace's head planes are used (gsx_raster_common.hpp)
    uint32_t tile_id;
    if (!swizzled_tile(blockIdx.x, a.tw * a.th, tile_id)) return;
    if (a.masks != nullptr && !a.masks[(size_t)cid * a.th * a.tw + tile_id]) return;  // Bwd.cu:84-86
    const uint32_t tile_y = tile_id / a.tw, tile_x = tile_id - tile_y * a.tw;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t i, j;
    thread_pixel(tid, tile_x, tile_y, i, j);
    const bool inside = i < a.H && j < a.W;
    const size_t pix = (size_t)cid * a.H * a.W + (size_t)min(i, a.H - 1) * a.W + min(j, a.W - 1);
    const float* bg = a.backgrounds ? a.backgrounds + cid * 3 : nullptr;
    if (KIND == CAM_OPENCV_FISHEYE && a.tile_flags != nullptr && a.tile_flags[(size_t)cid * a.th * a.tw + tile_id]) return;  // generic kernel's tile
    const Camera<KIND> cam(a.cams, cid, a.W, a.H);
    float u, v, w;
    const bool ray_ok = pixel_ray(cam, i, j, u, v, w);
    const bool active = inside && ray_ok;
    if (tid == 0) s_blockmax = -1;
    float wb[4], tb[4];
    if (KIND == CAM_OPENCV_FISHEYE) {
        const bool wide = active && w < 0.05f;
        const float iw = 1.f / fmaxf(w, 0.05f);
        uv_bounds(active, u * iw, v * iw, wave, lane, s_bounds, wb, tb, wide);   // contains a barrier
    } else {
        uv_bounds(active, u, v, wave, lane, s_bounds, wb, tb);   // contains a barrier
    }
    const float ww = w * w;
    const bool no_cull = KIND == CAM_OPENCV_FISHEYE && !(tb[0] > -INFINITY);

    int32_t range_start, range_end;
    tile_list_range(a, cid, tile_x, tile_y, range_start, range_end);

    const float T_final = 1.f - render_alphas[pix];
    float T = T_final;
    const int32_t bin_final = active ? last_ids[pix] : -1;
    const float vr = v_render_colors[pix * 3], vg = v_render_colors[pix * 3 + 1], vb = v_render_colors[pix * 3 + 2];
    const float va = v_render_alphas ? v_render_alphas[pix] : 0.f;
    float tail = va;  // T_final * ra * (v_alpha_out - bg . v_out)   (Bwd.cu:307-316)
    if (bg) tail -= bg[0] * vr + bg[1] * vg + bg[2] * vb;
    tail *= T_final;
    float tbuf = tail;  // tail - v . (colour buffer behind the current Gaussian)

    // moment whose total this lane's quad holds after butterfly_reduce16: 4 * quad + {0,2,1,3}[row]
    const uint32_t bf_row = lane >> 4;
    const uint32_t mom_of_lane = 4u * ((lane >> 2) & 3u) + ((bf_row == 1u) ? 2u : (bf_row == 2u ? 1u : bf_row));
    int32_t wave_last = bin_final;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wave_last = max(wave_last, __shfl_xor(wave_last, o));
    if (lane == 0) atomicMax(&s_blockmax, wave_last);
    __syncthreads();
    const int32_t block_last = min(s_blockmax, range_end - 1);
    if (block_last < range_start) return;
    const int32_t n_chunks = (block_last - range_start + BCH) / BCH;

    for (int32_t b = 0; b < n_chunks; ++b) {
        __syncthreads();  // previous chunk's records are written, LDS planes are free
        const int32_t chunk_end = block_last - BCH * b;  // inclusive; slot t holds sorted index chunk_end - t
        const int32_t chunk_size = min(BCH, chunk_end + 1 - range_start);
        if ((int32_t)tid < chunk_size) {
            const int32_t g = a.flatten_ids[chunk_end - (int32_t)tid];
            StagedRec sr;
            stage_one(a, tb, g, sr, tile_x, tile_y);
            if (no_cull) sr.cull.z = INFINITY;
            s_rec[tid][0] = sr.r0; s_rec[tid][1] = sr.r1; s_rec[tid][2] = sr.r2; s_rec[tid][3] = sr.r3;
            s_cull[tid] = sr.cull;
            s_gid[tid] = g;
        }
        if (tid < BCH) {
#pragma unroll
            for (int k = 0; k < NMOM; ++k) s_acc[k][tid] = 0.f;
        }
        if (tid < (BCH + 63) / 64) s_touched[tid] = 0ull;
        __syncthreads();

        for (int32_t sub = 0; sub < chunk_size; sub += 64) {
            bool hit = false;
            if (sub + (int32_t)lane < chunk_size && chunk_end - (sub + (int32_t)lane) <= wave_last) {
                const float4 c = s_cull[sub + lane];
                const float4 q0 = s_rec[sub + lane][0], q1 = s_rec[sub + lane][1];   // (.., .., l00, l01), (l11, ..)
                hit = footprint_hits(c, q0.z, q0.w, q1.x, wb[0], wb[1], wb[2], wb[3]);
            }
            unsigned long long todo = __builtin_amdgcn_ballot_w64(hit);
            unsigned long long touched = 0ull;
            while (todo) {
                const int t = sub + __builtin_ctzll(todo);
                todo &= todo - 1ull;
                const float4* rp = s_rec[t];
                const float4 r0 = rp[0], r1 = rp[1], r2 = rp[2], r3 = rp[3];
                float x0, x1, num2, rden;   // (x0, x1): whitened pixel offsets — the moments are taken in them (gsx_record.hpp: moments_to_gradients)
                const float alpha = KIND == CAM_OPENCV_FISHEYE ? fast_alpha_ray(u, v, w, ww, r0, r1, r2, x0, x1, num2, rden)
                                                               : fast_alpha(u, v, r0, r1, r2, x0, x1, num2, rden);
                // one comparison feeds the ballot directly (a ballot of `a && b` goes through a VGPR round trip)
                const float alpha_in = (chunk_end - t <= bin_final) ? alpha : 0.f;
                const bool valid = alpha_in >= ALPHA_MIN;
                if (__builtin_amdgcn_ballot_w64(valid) == 0ull) continue;
                touched |= 1ull << (t - sub);
                // branch-free: lanes that do not take this Gaussian run with alpha = 0, which makes every
                // update below the identity (ra = 1, fac = 0) and every moment weight zero.
                const float al = valid ? alpha_in : 0.f;
                const float ra = __builtin_amdgcn_rcpf(1.f - al);
                T *= ra;
                const float fac = al * T;
                float x[16];
                x[0] = fac * vr; x[1] = fac * vg; x[2] = fac * vb;
                // dL/dalpha = sum_c v_c (colour_c T - buffer_c ra) + tail ra  (Bwd.cu:286-316) only needs the scalars
                // cv = v . colour and tbuf = tail - v . buffer: one accumulator instead of the three colour buffers
                const float cv = fmaf(r3.y, vb, fmaf(r3.x, vg, r2.w * vr));
                const float v_alpha = fmaf(T, cv, ra * tbuf);
                tbuf = fmaf(-cv, fac, tbuf);
                // clamped alpha (>= 0.999) carries no gradient to opacity / geometry (Bwd.cu:318)
                const float av = (al < 0.999f) ? al * v_alpha : 0.f;   // o * v_opacity
                const float aw = av * rden;                            // -2 (dL/dD) / den'  (dalpha/dD = -alpha/2: the gather kernel applies the -1/2)
                const float bw = aw * (num2 * rden);                   // the same times D (0.5 log2 e)
                x[3] = av;
                x[7] = aw * x0; x[8] = aw * x1;                         // first-order moments, then the second-order ones from them
                x[4] = x[7] * x0; x[5] = x[7] * x1; x[6] = x[8] * x1;   // (10 products instead of 13 with x0^2, x0 x1, x1^2 formed first)
                x[9] = bw; x[10] = bw * x0; x[11] = bw * x1;
                x[12] = x[10] * x0; x[13] = x[10] * x1; x[14] = x[11] * x1;
                if (KIND == CAM_OPENCV_FISHEYE) {
                    // unnormalised rays: du' = u - w u0 has d/du0 = -w and A d = w h + a0 du' + a1 dv', so the moments that multiply
                    // d/du0, d/dv0 (first-order a) and h (the b family's 1, du', dv') carry the matching powers of w; the gather
                    // kernel's linear map is unchanged (the whitening is linear in (du', dv'))
                    x[7] *= w; x[8] *= w; x[9] *= ww; x[10] *= w; x[11] *= w;
                }
                x[15] = 0.f;
                const float total = butterfly_reduce16(x);
                if ((lane & 3u) == 0u) atomicAdd(&s_acc[mom_of_lane][t], total);  // 16 lanes, one moment each
            }
            if (lane == 0 && touched) atomicOr(&s_touched[sub >> 6], touched);
        }
        __syncthreads();

        // one thread per touched Gaussian of the chunk: 64 B moment record at its sorted index, chained per Gaussian
        if ((int32_t)tid < chunk_size && ((s_touched[tid >> 6] >> (tid & 63u)) & 1ull)) {
            int32_t prev;
            const int32_t isect = claim_record_slot(a, ws_head, ranges, s_gid[tid], chunk_end - (int32_t)tid, tile_x, tile_y, prev);
            float4* rec = ws_rec + (size_t)isect * 4;
            if ((int64_t)isect < a.rec_capacity) {   // (always, by construction: a run never exceeds the Gaussian's tile rectangle)
            nt_store4(make_float4(s_acc[0][tid], s_acc[1][tid], s_acc[2][tid], s_acc[3][tid]), rec);   // (nontemporal: see raster_bwd_gq_kernel)
            nt_store4(make_float4(s_acc[4][tid], s_acc[5][tid], s_acc[6][tid], s_acc[7][tid]), rec + 1);
            nt_store4(make_float4(s_acc[8][tid], s_acc[9][tid], s_acc[10][tid], s_acc[11][tid]), rec + 2);
            nt_store4(make_float4(s_acc[12][tid], s_acc[13][tid], s_acc[14][tid], __int_as_float(prev)), rec + 3);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward, Gaussian-major ("GQ"): lanes = Gaussians, loop over pixels
// ------------------------------------------------------------------------------------------------
// raster_bwd_fast_kernel above evaluates one Gaussian per step on the 64 pixels of a quadrant and pays, per step, a 16-value
// reduction over the wave (the butterfly, ~35 VALU) plus an LDS atomic — and 65-70 % of the lanes carry no weight, because a thin
// ellipse covers a fraction of an 8x8 block.  The Gaussian-major kernel transposes the loop: the tile's Gaussians are binned per
// 4x4 pixel block (the footprint of a thin ellipse fits a 4x4 block much better), Gaussians of a block's list sit on the lanes and
// the wave walks the block's pixels.  Per pixel every lane evaluates ITS Gaussian; the back-to-front recurrences
//     T_j = T_in * prod_{i<=j} 1/(1-alpha_i)            tbuf_j = tbuf_in - sum_{i<j} (c_i . v) alpha_i T_i
// become one multiplicative and one additive DPP scan over the lanes, and the 15 moments of a Gaussian are summed over the pixels in
// the lane's own registers.  Moment records / list heads / the gather kernel are those of the pixel-major kernel (same 15 moments,
// same layout).  First version (round 2, removed again): 64 Gaussians per batch on the 64 lanes, four row passes per block, scans of
// six steps — 0.72 ms at S-1M against 0.86 pixel-major, batches 73 % full.  Measured and simulated numbers: DESIGN.md §4.
#ifndef GSX_GS
#define GSX_GS 256
#endif
constexpr int GS = GSX_GS;   // Gaussians per super-chunk (one per thread at staging time; <= 256: list entries are bytes)

struct GmLaneRec { float u0, v0, l00, l01, l11, lo, d1, d2, d3, d4, d5, cr, cg, cb; int32_t idx; };
struct GmRowPix { float T[4], tb[4], vr[4], vg[4], vb[4]; int32_t binf[4]; };

// pixel owned by thread `tid` in the Gaussian-major kernel: wave = 8x8 quadrant, DPP row (16 lanes) = 4x4 block of the quadrant
GSX_DEV void thread_pixel_gm(uint32_t tid, uint32_t tile_x, uint32_t tile_y, uint32_t& i, uint32_t& j) {
    const uint32_t wave = tid >> 6, sb = (tid >> 4) & 3u, p = tid & 15u;
    j = tile_x * TILE + (wave & 1u) * 8u + (sb & 1u) * 4u + (p & 3u);
    i = tile_y * TILE + (wave >> 1) * 8u + (sb >> 1) * 4u + (p >> 2);
}

#ifndef GSX_GM_WAVES
#define GSX_GM_WAVES 4
#endif
#ifndef GSX_GQ_SLEEP
#define GSX_GQ_SLEEP 1   // s_sleep argument while waiting for the tile's flush lock
#endif

// ---- batch quantum 16: lanes = 16 Gaussians x the 4 pixel rows of a 4x4 block --------------------------------------------------
// With 64 Gaussians of a block's list on the 64 lanes, a (wave, block) that sees ~35-57 Gaussians per super-chunk (S-1M) runs its
// batch 73 % full.  Here a DPP row (16 lanes) holds 16 Gaussians of the
// list and the wave's four DPP rows are the block's four pixel ROWS: one pass of the (same) 4-pixel body covers the whole 4x4 block
// for 16 Gaussians, lists are consumed 16 at a time (fill ~0.9) and both scans stay inside a DPP row (row_shr 1/2/4/8: four steps
// instead of six, no row_bcast).  What it costs: the per-pixel inputs differ between the rows, so they live in VGPRs (loaded once
// per (wave, block, super-chunk)) instead of SGPRs; the carries T / tbuf travel from lane 15 of each row to the row's next pass with
// a ds_swizzle broadcast (LDS crossbar, no VALU); and the four rows' partial moments of a Gaussian are summed with the first two
// stages of the halving butterfly (v_permlane32_swap, v_permlane16_swap: 24 VALU per pass) before ONE lock-protected read-add-write
// of four values per lane.  Super-chunks are balanced (a tile with 412 Gaussians stages 2 x 206, not 256 + 156).
#define GSX_SCANR_STEP(OP, CTRL) OP " %0, %0, %0 " CTRL "\n\t" OP " %1, %1, %1 " CTRL "\n\t" OP " %2, %2, %2 " CTRL "\n\t" OP " %3, %3, %3 " CTRL "\n\t"
#define GSX_SCANR(OP)                                                      \
    asm volatile("s_nop 1\n\t"                                             \
                 GSX_SCANR_STEP(OP, "row_shr:1 row_mask:0xf bank_mask:0xf")    \
                 GSX_SCANR_STEP(OP, "row_shr:2 row_mask:0xf bank_mask:0xf")    \
                 GSX_SCANR_STEP(OP, "row_shr:4 row_mask:0xf bank_mask:0xf")    \
                 GSX_SCANR_STEP(OP, "row_shr:8 row_mask:0xf bank_mask:0xf")    \
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]))
GSX_DEV void row_scan4_mul(float (&x)[4]) { GSX_SCANR("v_mul_f32_dpp"); }
GSX_DEV void row_scan4_add(float (&x)[4]) { GSX_SCANR("v_add_f32_dpp"); }
#undef GSX_SCANR
// The additive scan into OTHER registers: its first step reads the summands through DPP with bound_ctrl (a lane without a neighbour reads 0) and
// writes the sum elsewhere, so the summands survive without four v_mov (they are needed again behind the scan).  Not for the product (0 is not its identity).
GSX_DEV void row_scan4_add_to(const float (&e)[4], float (&x)[4]) {
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "v_add_f32_dpp %1, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "v_add_f32_dpp %2, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "v_add_f32_dpp %3, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "s_nop 1\n\t"
                 GSX_SCANR_STEP("v_add_f32_dpp", "row_shr:2 row_mask:0xf bank_mask:0xf")
                 GSX_SCANR_STEP("v_add_f32_dpp", "row_shr:4 row_mask:0xf bank_mask:0xf")
                 GSX_SCANR_STEP("v_add_f32_dpp", "row_shr:8 row_mask:0xf bank_mask:0xf")
                 : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3])
                 : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]));
}
#undef GSX_SCANR_STEP

// lane 15 of every DPP row -> all 16 lanes of that row (ds_swizzle bit mode: lane' = (lane & 0x10) | 0x0f inside each half)
GSX_DEV float row_last(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, x), 0x01F0)); }

// Sums x[0..15] over the four DPP rows (lanes l, l+16, l+32, l+48).  Afterwards z[j] of a lane in row r is the total of value
// 4 j + {0,2,1,3}[r] for the lane's column (the first two stages of butterfly_reduce16).
// (Round 5: the same two stages through the LDS crossbar — ds_bpermute for lane ^ 32, ds_swizzle for lane ^ 16, two selects + an add per pair instead of
// a quarter-rate swap + an add: 246 VALU but ~46 fewer issue cycles per pass on paper — measured 0.5309 -> 0.5431 ms same-box: the LDS is the co-limit.)
GSX_DEV void rows_reduce16(float (&x)[16], float (&z)[4]) {
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
#pragma unroll
    for (int j = 0; j < 4; ++j) z[j] = y[2 * j] + y[2 * j + 1];
}

// One row (4 pixels) of a 4x4 block for the 16 Gaussians of a DPP row.  CLAMP = false when no Gaussian of the pass can reach alpha
// 0.999 (opacity < 0.999: alpha = o exp(-s) stays below it), which drops the clamp and its gradient mask.
// ROWDV: dv is the same for the lane's four pixels (perfect pinhole: a lane's pixels are one image row), so the dv factors of the
// moments are applied once per pass to three sums per weight instead of per pixel (12 instead of 19 accumulation VALU per pixel).
// FISH: unnormalised rays (u, v, w): du' = u - w u0 has d/du0 = -w and A d = w h + a0 du' + a1 dv', so the moments that multiply d/du0,
// d/dv0 (first-order a) and h (the b family's 1, du', dv') carry the matching powers of the pixel's w (as in raster_bwd_fast_kernel).
template <bool CLAMP, bool ROWDV, bool FISH>
GSX_DEV void gq_row(const GmLaneRec& g, const GmRowPix& px, const float (&x0)[4], const float (&x1)[4], const float (&pw)[4], const float (&num2)[4],
                    const float (&rden)[4], float st, float st2, float (&acc)[16], float (&T_out)[4], float (&tb_out)[4]) {
    float al[4], ra[4], P[4];
    // (opaque copy: the clamped and the clamp-free instantiation of this function sit in the two arms of one branch, and hipcc otherwise hoists the
    // four `idx <= last id` compares above it for the clamped arm AND recomputes them in the clamp-free one — 4 of what were 255 VALU per pass)
    int32_t idx = g.idx;
    asm volatile("" : "+v"(idx));
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        float alpha = __builtin_amdgcn_exp2f(fmaf(-num2[h], rden[h], g.lo));
        if (CLAMP) alpha = fminf(0.999f, alpha);
        const bool valid = (idx <= px.binf[h]) && (alpha >= ALPHA_MIN);
        al[h] = valid ? alpha : 0.f;
        P[h] = ra[h] = __builtin_amdgcn_rcpf(1.f - al[h]);   // (round 5: the four reciprocals as one asm run of v_rcp_f32 — A/B 0.5421 -> 0.5449 ms, not kept)
    }
    row_scan4_mul(P);
    float T[4], fac[4], cv[4], e[4], S[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        T[h] = px.T[h] * P[h];                                   // transmittance in front of this Gaussian
        fac[h] = al[h] * T[h];
        cv[h] = fmaf(g.cb, px.vb[h], fmaf(g.cg, px.vg[h], g.cr * px.vr[h]));
        e[h] = cv[h] * fac[h];
    }
    row_scan4_add_to(e, S);
    float awh[4], bwh[4];
    (void)awh; (void)bwh;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const float tbo = px.tb[h] - S[h];                       // what the NEXT Gaussian of the row starts from (S = inclusive sum of e over the row's lanes)
        // alpha dL/dalpha = alpha (T c.v + tbuf / (1 - alpha)) with tbuf = tbo + e (tail - v . colour accumulated behind this Gaussian) and e = c.v alpha T:
        //                 = e (1 + alpha / (1 - alpha)) + (alpha / (1 - alpha)) tbo = e ra + (alpha ra) tbo        (1 + alpha / (1 - alpha) = ra)
        float av = fmaf(al[h] * ra[h], tbo, e[h] * ra[h]);
        if (CLAMP) av = (al[h] < 0.999f) ? av : 0.f;             // clamped alpha carries no gradient (Bwd.cu:318)
        const float aw = av * rden[h];
        const float bw = aw * (num2[h] * rden[h]);
        acc[0] = fmaf(fac[h], px.vr[h], acc[0]); acc[1] = fmaf(fac[h], px.vg[h], acc[1]); acc[2] = fmaf(fac[h], px.vb[h], acc[2]);
        acc[3] += av;
        if (ROWDV) {   // acc[6] / acc[8] / acc[11] collect sum aw, sum aw (again), sum bw here; the dv factors follow below
#ifdef GSX_GQ_NO_HMOMENTS
            const float x7 = aw * x0[h], x10 = bw * x0[h];
            acc[4] = fmaf(x7, x0[h], acc[4]); acc[7] += x7; acc[8] += aw;
            acc[9] += bw; acc[10] += x10; acc[12] = fmaf(x10, x0[h], acc[12]);
#else
            awh[h] = aw; bwh[h] = bw;   // the x0 moments of the lane's four pixels are formed after the loop from moments in h (x0[h] = x0[0] + h st)
#endif
        } else {
            const float x7 = aw * x0[h], x8 = aw * x1[h], x10 = bw * x0[h], x11 = bw * x1[h];
            acc[4] = fmaf(x7, x0[h], acc[4]); acc[5] = fmaf(x7, x1[h], acc[5]); acc[6] = fmaf(x8, x1[h], acc[6]);
            if (FISH) {
                acc[7] = fmaf(x7, pw[h], acc[7]); acc[8] = fmaf(x8, pw[h], acc[8]); acc[9] = fmaf(bw, pw[h] * pw[h], acc[9]);
                acc[10] = fmaf(x10, pw[h], acc[10]); acc[11] = fmaf(x11, pw[h], acc[11]);
            } else {
                acc[7] += x7; acc[8] += x8; acc[9] += bw; acc[10] += x10; acc[11] += x11;
            }
            acc[12] = fmaf(x10, x0[h], acc[12]); acc[13] = fmaf(x10, x1[h], acc[13]); acc[14] = fmaf(x11, x1[h], acc[14]);
        }
        T_out[h] = T[h]; tb_out[h] = tbo;
    }
    if (ROWDV) {
#ifndef GSX_GQ_NO_HMOMENTS
        // A lane's four pixels are one image row: du[h] = du0 + h / fx, x1 is theirs in common, so x0[h] = x0[0] + h st with st = l00 / fx.  With
        // S0 = sum w, S1 = sum h w, S2 = sum h^2 w (h = 0 .. 3: constants)
        //     sum w x0 = x0[0] S0 + st S1,     sum w x0^2 = x0[0] (sum w x0 + st S1) + st^2 S2
        // 12 VALU per weight family instead of 16 (three per pixel + the weight's own sum), no per-pixel products with x0.
        {
            const float xa = x0[0];
            const float S0a = (awh[0] + awh[1]) + (awh[2] + awh[3]), S1a = fmaf(3.f, awh[3], fmaf(2.f, awh[2], awh[1])), S2a = fmaf(9.f, awh[3], fmaf(4.f, awh[2], awh[1]));
            const float S0b = (bwh[0] + bwh[1]) + (bwh[2] + bwh[3]), S1b = fmaf(3.f, bwh[3], fmaf(2.f, bwh[2], bwh[1])), S2b = fmaf(9.f, bwh[3], fmaf(4.f, bwh[2], bwh[1]));
            const float Aa = st * S1a, Ab = st * S1b;
            acc[8] = S0a; acc[7] = fmaf(xa, S0a, Aa); acc[4] = fmaf(xa, acc[7] + Aa, st2 * S2a);
            acc[9] = S0b; acc[10] = fmaf(xa, S0b, Ab); acc[12] = fmaf(xa, acc[10] + Ab, st2 * S2b);
        }
#endif
        const float d = x1[0], a0 = acc[8] * d, b0 = acc[9] * d;   // sum aw x1, sum bw x1
        acc[5] = acc[7] * d; acc[6] = a0 * d; acc[8] = a0;
        acc[11] = b0; acc[13] = acc[10] * d; acc[14] = b0 * d;
    }
}

template <int KIND>
__global__ __launch_bounds__(RB, GSX_GM_WAVES) void raster_bwd_gq_kernel(RasterArgs a, const float* __restrict__ render_alphas,
                                                                          const int32_t* __restrict__ last_ids,
                                                                          const float* __restrict__ v_render_colors,
                                                                          const float* __restrict__ v_render_alphas,
                                                                          float4* __restrict__ ws_rec, int32_t* __restrict__ ws_head) {
    // record planes: 0 u0, 1 v0, 2 l00, 3 l01, 4 l11, 5 lo, 6 d1, 7 d2, 8 d3, 9 d4, 10 d5, 11 red, 12 green, 13 blue, 14 rad2, 15 k2 (footprint())
    __shared__ float s_rec[16][GS];
    __shared__ float s_acc[16][GS + 8];      // (+8: the four pixel rows of a pass flush the same slot to planes p, p+1, p+2, p+3 -> four different banks) plane 15 (the butterfly's padding value) counts the passes that listed the slot: > 0 = touched
    __shared__ int32_t s_gid[GS];
    __shared__ uint8_t s_list[16][GS];       // [wave * 4 + k]: slots of the super-chunk whose footprint reaches 4x4 block k of the wave's quadrant
    __shared__ float s_T[RB], s_tbuf[RB];    // per pixel: the two loop-carried quantities of the back-to-front recurrence, between super-chunks
    __shared__ uint32_t s_lock;
    __shared__ float4 s_uvb[KIND == CAM_PERFECT_PINHOLE ? 1 : RB];   // distorted cameras: per pixel (u, v, last id, w) (w = 1 unless fisheye)
    __shared__ float s_bounds[4][4];
    __shared__ int32_t s_blockmax;
    GSX_BLOCK_CLOCK(2);
    const uint32_t cid = blockIdx.y;
    const bool ranges = record_ranges(a, ws_head);   // how this workspace's head planes are used (gsx_raster_common.hpp)
    uint32_t tile_id;
    if (!swizzled_tile(blockIdx.x, a.tw * a.th, tile_id)) return;
    if (a.masks != nullptr && !a.masks[(size_t)cid * a.th * a.tw + tile_id]) return;  // Bwd.cu:84-86
    const uint32_t tile_y = tile_id / a.tw, tile_x = tile_id - tile_y * a.tw;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t uwave = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);   // the same number, known to be wave-uniform
    uint32_t i, j;
    thread_pixel_gm(tid, tile_x, tile_y, i, j);
    const bool inside = i < a.H && j < a.W;
    const size_t pix = (size_t)cid * a.H * a.W + (size_t)min(i, a.H - 1) * a.W + min(j, a.W - 1);
    const float* bg = a.backgrounds ? a.backgrounds + cid * 3 : nullptr;
    if (KIND == CAM_OPENCV_FISHEYE && a.tile_flags != nullptr && a.tile_flags[(size_t)cid * a.th * a.tw + tile_id]) return;  // generic kernel's tile
    const Camera<KIND> cam(a.cams, cid, a.W, a.H);
    float u, v, w;
    const bool ray_ok = pixel_ray(cam, i, j, u, v, w);
    const bool active = inside && ray_ok;
    if (tid == 0) { s_blockmax = -1; s_lock = 0u; }
    float wb[4], tb[4];
    // footprints live in the chart (u / w, v / w); a fisheye pixel near or beyond 90 degrees has none: its block's / wave's / tile's bounds
    // become infinite, which switches the culling off for them
    const bool wide = KIND == CAM_OPENCV_FISHEYE && active && w < 0.05f;
    const float iw = KIND == CAM_OPENCV_FISHEYE ? 1.f / fmaxf(w, 0.05f) : 1.f;
    const float uc = u * iw, vc = v * iw;
    uv_bounds(active, uc, vc, wave, lane, s_bounds, wb, tb, wide);   // contains a barrier
    const bool no_cull = KIND == CAM_OPENCV_FISHEYE && !(tb[0] > -INFINITY);

    int32_t range_start, range_end;
    tile_list_range(a, cid, tile_x, tile_y, range_start, range_end);

    {   // per-pixel carries -> LDS
        const float T_final = 1.f - render_alphas[pix];
        float tail = v_render_alphas ? v_render_alphas[pix] : 0.f;   // T_final * (v_alpha_out - bg . v_out)   (Bwd.cu:307-316)
        if (bg) tail -= bg[0] * v_render_colors[pix * 3] + bg[1] * v_render_colors[pix * 3 + 1] + bg[2] * v_render_colors[pix * 3 + 2];
        s_T[tid] = T_final; s_tbuf[tid] = tail * T_final;
        if (KIND != CAM_PERFECT_PINHOLE) s_uvb[tid] = make_float4(u, v, __int_as_float(active ? last_ids[pix] : -1), w);
    }
    // bounds and last id of this lane's 4x4 block (row of 16 lanes), then as wave-uniform scalars per block
    float bu0 = wide ? -INFINITY : (active ? uc : INFINITY), bu1 = wide ? INFINITY : (active ? uc : -INFINITY);
    float bv0 = wide ? -INFINITY : (active ? vc : INFINITY), bv1 = wide ? INFINITY : (active ? vc : -INFINITY);
    int32_t blast = active ? last_ids[pix] : -1;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        bu0 = fminf(bu0, __shfl_xor(bu0, o)); bu1 = fmaxf(bu1, __shfl_xor(bu1, o));
        bv0 = fminf(bv0, __shfl_xor(bv0, o)); bv1 = fmaxf(bv1, __shfl_xor(bv1, o));
        blast = max(blast, __shfl_xor(blast, o));
    }
    float sbb[4][4];
    int32_t sb_last[4];
#pragma unroll
    for (int sb = 0; sb < 4; ++sb) {
        sbb[sb][0] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bu0), sb * 16));
        sbb[sb][1] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bu1), sb * 16));
        sbb[sb][2] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bv0), sb * 16));
        sbb[sb][3] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bv1), sb * 16));
        sb_last[sb] = __builtin_amdgcn_readlane(blast, sb * 16);
    }
    const int32_t wave_last = max(max(sb_last[0], sb_last[1]), max(sb_last[2], sb_last[3]));
    if (lane == 0) atomicMax(&s_blockmax, wave_last);
    __syncthreads();
    const int32_t block_last = min(s_blockmax, range_end - 1);
    if (block_last < range_start) return;
    const int32_t n_total = block_last - range_start + 1;
    const int32_t n_super = (n_total + GS - 1) / GS;
    const int32_t per_super = (n_total + n_super - 1) / n_super;   // balanced super-chunks (<= GS)

    const uint32_t prow = lane >> 4, pcol = lane & 15u;            // this lane's pixel row of the block / its Gaussian column of the batch
    const float su = 1.f / cam.fx, sv = 1.f / cam.fy;
    const float prow_f = (float)prow;
    // plane of value z[j] after rows_reduce16: 4 j + {0,2,1,3}[row]
    const uint32_t zplane0 = (prow == 1u) ? 2u : (prow == 2u ? 1u : prow);

    for (int32_t sc = 0; sc < n_super; ++sc) {
        __syncthreads();  // previous super-chunk's records are written, LDS planes are free
        const int32_t chunk_end = block_last - per_super * sc;  // inclusive; slot t holds sorted index chunk_end - t (back to front)
        const int32_t chunk_size = min(per_super, chunk_end + 1 - range_start);
        if ((int32_t)tid < chunk_size) {
            const int32_t g = a.flatten_ids[chunk_end - (int32_t)tid];
            StagedRec sr;
            stage_one(a, tb, g, sr, tile_x, tile_y);
            s_rec[0][tid] = sr.r0.x; s_rec[1][tid] = sr.r0.y; s_rec[2][tid] = sr.r0.z; s_rec[3][tid] = sr.r0.w;
            s_rec[4][tid] = sr.r1.x; s_rec[5][tid] = sr.r1.y; s_rec[6][tid] = sr.r1.z; s_rec[7][tid] = sr.r1.w;
            s_rec[8][tid] = sr.r2.x; s_rec[9][tid] = sr.r2.y; s_rec[10][tid] = sr.r2.z; s_rec[11][tid] = sr.r2.w;
            s_rec[12][tid] = sr.r3.x; s_rec[13][tid] = sr.r3.y; s_rec[14][tid] = no_cull ? INFINITY : sr.cull.z; s_rec[15][tid] = sr.cull.w;
            s_gid[tid] = g;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) s_acc[k][tid] = 0.f;
        __syncthreads();

        if (wave == 0) { GSX_STAT_ADD(10, 1); GSX_STAT_ADD(11, chunk_size); }
        // ---- bin the super-chunk's Gaussians into the lists of this wave's four 4x4 blocks (back-to-front order is kept) ----
        uint32_t cnt[4] = {0u, 0u, 0u, 0u};
        for (int32_t c0 = 0; c0 < chunk_size; c0 += 64) {
            const int32_t c = c0 + (int32_t)lane;
            const bool in = c < chunk_size;
            const float4 cc = make_float4(s_rec[0][c & (GS - 1)], s_rec[1][c & (GS - 1)], s_rec[14][c & (GS - 1)], s_rec[15][c & (GS - 1)]);
            const float c00 = s_rec[2][c & (GS - 1)], c01 = s_rec[3][c & (GS - 1)], c11 = s_rec[4][c & (GS - 1)];
            const int32_t idx = chunk_end - c;
            uint32_t hits4 = 0u;
            if (KIND == CAM_PERFECT_PINHOLE) {   // the blocks' (u, v) rectangles are products of two u-ranges and two v-ranges: shared 1-D pieces
                const float xr[2][2] = {{sbb[0][0], sbb[0][1]}, {sbb[1][0], sbb[1][1]}}, yr[2][2] = {{sbb[0][2], sbb[0][3]}, {sbb[2][2], sbb[2][3]}};
                hits4 = footprint_hits_2x2(cc, c00, c01, c11, xr, yr);
            }
#pragma unroll
            for (int sb = 0; sb < 4; ++sb) {
                const bool fp = KIND == CAM_PERFECT_PINHOLE ? ((hits4 >> sb) & 1u) != 0u
                                                             : footprint_hits(cc, c00, c01, c11, sbb[sb][0], sbb[sb][1], sbb[sb][2], sbb[sb][3]);
                const bool hit = in && idx <= sb_last[sb] && fp;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
                if (hit) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    s_list[wave * 4 + sb][cnt[sb] + rank] = (uint8_t)c;
                }
                cnt[sb] += (uint32_t)__popcll(m);
            }
        }
        __builtin_amdgcn_wave_barrier();

#pragma unroll 1
        for (int sb = 0; sb < 4; ++sb) {
            const uint32_t n_list = cnt[sb];
            if (n_list == 0u) continue;
            GSX_STAT_ADD(8, n_list); GSX_STAT_ADD(9, (n_list + 15u) / 16u);
            const uint8_t* list = s_list[wave * 4 + sb];
            // ---- this lane's four pixels: row `prow` of block sb, columns 0..3 (inputs in VGPRs for the whole list) ----
            const uint32_t bx = tile_x * TILE + (uwave & 1u) * 8u + ((uint32_t)sb & 1u) * 4u, by = tile_y * TILE + (uwave >> 1) * 8u + ((uint32_t)sb >> 1) * 4u;
            const uint32_t y = by + prow;
            const uint32_t cbase = wave * 64u + (uint32_t)sb * 16u + prow * 4u;   // carries / s_uvb index of this lane's first pixel
            GmRowPix px;
            float pu[4], pv[4], pw[4];
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const uint32_t x = bx + (uint32_t)h;
                const size_t gp = (size_t)cid * a.H * a.W + (size_t)min(y, a.H - 1) * a.W + min(x, a.W - 1);
                px.vr[h] = v_render_colors[gp * 3]; px.vg[h] = v_render_colors[gp * 3 + 1]; px.vb[h] = v_render_colors[gp * 3 + 2];
                px.T[h] = s_T[cbase + h]; px.tb[h] = s_tbuf[cbase + h];
                if (KIND == CAM_PERFECT_PINHOLE) {
                    px.binf[h] = (y < a.H && x < a.W) ? last_ids[gp] : -1;
                } else {
                    const float4 q = s_uvb[cbase + h];
                    px.binf[h] = __float_as_int(q.z);
                    pu[h] = q.x; pv[h] = q.y; pw[h] = q.w;
                }
            }
            const float bu = ((float)bx + 0.5f - cam.cx) * su;                       // perfect pinhole: u of the block's column 0
            const float pvr = fmaf(prow_f, sv, ((float)by + 0.5f - cam.cy) * sv);    //                  v of this lane's row
#pragma unroll 1
            for (uint32_t b0 = 0; b0 < n_list; b0 += 16) {
                const bool have = b0 + pcol < n_list;
                // (slot 0 for idle lanes, not the stale byte behind the list's end: an idle lane's alpha is 0, but its record still enters
                // cv = colour . v_out and the additive scan — a stale slot beyond the staged records holds uninitialised LDS, NaN * 0 = NaN; tried in round 5)
                const uint32_t slot = have ? (uint32_t)list[b0 + pcol] : 0u;
                GmLaneRec g;
                g.idx = chunk_end - (int32_t)slot;
                g.u0 = s_rec[0][slot]; g.v0 = s_rec[1][slot]; g.l00 = s_rec[2][slot]; g.l01 = s_rec[3][slot]; g.l11 = s_rec[4][slot];
                g.lo = have ? s_rec[5][slot] : -INFINITY;   // idle lanes: alpha = 0
                g.d1 = s_rec[6][slot]; g.d2 = s_rec[7][slot]; g.d3 = s_rec[8][slot]; g.d4 = s_rec[9][slot]; g.d5 = s_rec[10][slot];
                g.cr = s_rec[11][slot]; g.cg = s_rec[12][slot]; g.cb = s_rec[13][slot];
                const bool clamp = __builtin_amdgcn_ballot_w64(g.lo > -0.0015f) != 0ull;
                float x0[4], x1[4], num2[4], rden[4];   // (x0, x1) = L (du, dv): the whitened offsets, in which the moments are taken
                float st = 0.f, st2 = 0.f;              // perfect pinhole: x0[h + 1] - x0[h] = l00 / fx and its square
                if (KIND == CAM_PERFECT_PINHOLE) {
                    const float dvr = pvr - g.v0, du0 = bu - g.u0;
                    const float t1 = g.l11 * dvr, t1sq = t1 * t1, t0r = g.l01 * dvr;
                    const float Ar = fmaf(dvr, fmaf(g.d5, dvr, g.d2), 1.f), Br = fmaf(g.d4, dvr, g.d1);
                    st = g.l00 * su; st2 = st * st;
#pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        const float du = h == 0 ? du0 : fmaf((float)h, su, du0);   // (h == 0 spelled out: hipcc emits fma(0, su, du0) otherwise)
                        x0[h] = fmaf(g.l00, du, t0r); x1[h] = t1;
                        num2[h] = fmaf(x0[h], x0[h], t1sq);
                        rden[h] = __builtin_amdgcn_rcpf(fmaf(du, fmaf(g.d3, du, Br), Ar));
                    }
                } else if (KIND == CAM_OPENCV_FISHEYE) {   // unnormalised rays (u, v, w): fast_alpha_ray
#pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        const float du = fmaf(-pw[h], g.u0, pu[h]), dv = fmaf(-pw[h], g.v0, pv[h]);
                        x0[h] = fmaf(g.l01, dv, g.l00 * du);
                        x1[h] = g.l11 * dv;
                        num2[h] = fmaf(x0[h], x0[h], x1[h] * x1[h]);
                        rden[h] = __builtin_amdgcn_rcpf(fmaf(du, fmaf(g.d3, du, fmaf(g.d4, dv, g.d1 * pw[h])),
                                                             fmaf(dv, fmaf(g.d5, dv, g.d2 * pw[h]), pw[h] * pw[h])));
                    }
                } else {
#pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        const float du = pu[h] - g.u0, dv = pv[h] - g.v0;
                        x0[h] = fmaf(g.l01, dv, g.l00 * du);
                        x1[h] = g.l11 * dv;
                        num2[h] = fmaf(x0[h], x0[h], x1[h] * x1[h]);
                        rden[h] = __builtin_amdgcn_rcpf(fmaf(du, fmaf(g.d3, du, fmaf(g.d4, dv, g.d1)), fmaf(dv, fmaf(g.d5, dv, g.d2), 1.f)));
                    }
                }
                float acc[16], T_out[4], tb_out[4];
                if (clamp) gq_row<true, KIND == CAM_PERFECT_PINHOLE, KIND == CAM_OPENCV_FISHEYE>(g, px, x0, x1, pw, num2, rden, st, st2, acc, T_out, tb_out);
                else gq_row<false, KIND == CAM_PERFECT_PINHOLE, KIND == CAM_OPENCV_FISHEYE>(g, px, x0, x1, pw, num2, rden, st, st2, acc, T_out, tb_out);
                // (round 5: instantiating the rest of the pass in both arms, so that their 24 results need not meet in the same registers, was tried
                // for the clamp-free arm's four v_mov: 245 instead of 237 VALU per pass — the arms then disagree about more, not less)
                acc[15] = 1.f;   // "listed" marker (summed like a moment: no separate LDS atomic)
                // carries for this row's next pass: the values behind the row's last Gaussian
                // (round 5: the eight broadcasts as `v_mov_b32_dpp ... row_newbcast:15` instead of ds_swizzle — 8 LDS-crossbar operations fewer, 8 DPP VALU more:
                // measured 0.5367 -> 0.5400 ms same-box, not kept)
#pragma unroll
                for (int h = 0; h < 4; ++h) { px.T[h] = row_last(T_out[h]); px.tb[h] = row_last(tb_out[h]); }
                // the four pixel rows' partial moments of each Gaussian -> one total per (Gaussian, moment), four moments per lane
                float z[4];
                rows_reduce16(acc, z);
                // -> the tile's accumulator: inside the wave every (moment, slot) has one owner lane; the four waves exclude each other
                if (lane == 0u) {
                    uint32_t expected = 0u;
                    while (!__hip_atomic_compare_exchange_strong(&s_lock, &expected, 1u, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                        expected = 0u;
                        GSX_STAT_ADD(12, 1);   // (lane 0 is the only lane here)
                        __builtin_amdgcn_s_sleep(GSX_GQ_SLEEP);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                if (have) {
                    float cur[4];
#pragma unroll
                    for (int jz = 0; jz < 4; ++jz) cur[jz] = s_acc[4 * jz + zplane0][slot];
#pragma unroll
                    for (int jz = 0; jz < 4; ++jz) s_acc[4 * jz + zplane0][slot] = cur[jz] + z[jz];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0u) __hip_atomic_store(&s_lock, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            // carries of the block's pixels for the next super-chunk (every lane of a row holds the row's values)
            if (pcol == 0u) {
#pragma unroll
                for (int h = 0; h < 4; ++h) { s_T[cbase + h] = px.T[h]; s_tbuf[cbase + h] = px.tb[h]; }
            }
        }
        __syncthreads();

        // the touched Gaussians of the super-chunk leave as 64 B moment records (at their sorted index / in their Gaussian's run, chained per
        // Gaussian): one thread per record, four 16 B nontemporal stores.  The four quarters of a line reach HBM separately (WRITE_SIZE 436 MB
        // for 215 MB of records at S-1M, profiles/r04_pmc_counters.md).  GSX_GQ_REC_4LANE (round 5, measured and not the default): four lanes
        // per record, one quarter each, so that a wave instruction stores 16 whole lines — same-box A/B of the op 0.551 -> 0.568 ms (+3 %):
        // the stores are not what the kernel waits for, the extra LDS reads / shuffles and 8 more VGPRs are paid on its critical path.
#ifdef GSX_GQ_REC_4LANE
        for (int32_t base = 0; base < chunk_size; base += RB / 4) {
            const int32_t r = base + (int32_t)(tid >> 2);
            const uint32_t q = tid & 3u;
            const bool touched = r < chunk_size && s_acc[15][r & (GS - 1)] > 0.f;
            int32_t prev = -1, isect = 0;
            if (touched && q == 0u) isect = claim_record_slot(a, ws_head, ranges, s_gid[r], chunk_end - r, tile_x, tile_y, prev);
            isect = __shfl(isect, (int)(lane & ~3u));   // the quad's first lane claimed the slot
            prev = __shfl(prev, (int)(lane & ~3u));
            if (touched && (int64_t)isect < a.rec_capacity) {   // (always, by construction: a run never exceeds the Gaussian's tile rectangle)
                const float w3 = s_acc[4 * q + 3][r];            // (plane 15 is the pass counter: the record's last word is the chain link)
                // nontemporal: 64 B written once at a scattered slot and read once by the gather kernel — as ordinary stores the records
                // push the kernel's own working set (lists, packed records, pixel inputs) out of L2: S-1M 0.573 -> 0.526 ms, S-5M @4K 2.17 ->
                // 2.03, garden stand-in 1.28 -> 1.20 (same-box A/B; nontemporal LOADS in the gather measured flat)
                nt_store4(make_float4(s_acc[4 * q][r], s_acc[4 * q + 1][r], s_acc[4 * q + 2][r], q == 3u ? __int_as_float(prev) : w3), ws_rec + (size_t)isect * 4 + q);
            }
        }
#else
        if ((int32_t)tid < chunk_size && s_acc[15][tid] > 0.f) {
            int32_t prev;
            const int32_t isect = claim_record_slot(a, ws_head, ranges, s_gid[tid], chunk_end - (int32_t)tid, tile_x, tile_y, prev);
            float4* rec = ws_rec + (size_t)isect * 4;
            if ((int64_t)isect < a.rec_capacity) {
                nt_store4(make_float4(s_acc[0][tid], s_acc[1][tid], s_acc[2][tid], s_acc[3][tid]), rec);
                nt_store4(make_float4(s_acc[4][tid], s_acc[5][tid], s_acc[6][tid], s_acc[7][tid]), rec + 1);
                nt_store4(make_float4(s_acc[8][tid], s_acc[9][tid], s_acc[10][tid], s_acc[11][tid]), rec + 2);
                nt_store4(make_float4(s_acc[12][tid], s_acc[13][tid], s_acc[14][tid], __int_as_float(prev)), rec + 3);
            }
        }
#endif
    }
}

// Sum the moment records of every (camera, Gaussian) (lists built by raster_bwd_fast_kernel) and apply the chain
// rule once: moments -> (B0, B1, h, a_i, u0, v0, m_z) -> (A, m) -> (M, mu) -> (quat, scale) (Utils.cuh:104-158).
// One thread per Gaussian; colours / opacities are per camera, means / quats / scales are shared by the cameras.
// (111 VGPRs, 4 waves per SIMD — set by the chain rule, not by the walk.  Holding the allocation to 5 / 6 / 8 waves (launch bounds; 52 / 128 /
// 184 B of scratch per lane) measured +0.010 / +0.026 / +0.054 ms on the S-1M step: the kernel moves ~240 MB — 64 B records at scattered
// slots — in ~66 us and is near what such a read pattern gets from HBM, more chains in flight do not help it.)
// ACT (round 6): the activation Jacobians as the epilogue (ActEpilogue, gsx_raster_common.hpp; C == 1): v_quats / v_scales / v_opacities are not written.
// (the walk hides its 64 B pointer chases behind other waves: the kernel must stay at 4 waves / SIMD, <= 128 VGPRs — at 132, three waves, the S-1M gather
// went from 69 to 79 us; a forced bound spills and costs the same.  -Rpass-analysis=kernel-resource-usage after every change to this kernel.)
template <int KIND, int NCH, bool ACT>
__global__ __launch_bounds__(256) void gsx_bwd_gather_kernel(RasterArgs a, const float4* __restrict__ ws_rec,
                                                             int32_t* __restrict__ ws_head, float* __restrict__ v_means,
                                                             float* __restrict__ v_quats, float* __restrict__ v_scales,
                                                             float* __restrict__ v_colors, float* __restrict__ v_opacities, ActEpilogue act) {
    const uint32_t gi_raw = blockIdx.x * 256u + threadIdx.x;
    const bool in = gi_raw < a.N;                       // (no early return: the long runs of the range mode are summed by the whole wave)
    const uint32_t gi = in ? gi_raw : a.N - 1u;
    const uint32_t lane = threadIdx.x & 63u;
    const bool ranges = record_ranges(a, ws_head);
    float geo[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) geo[k] = 0.f;
    bool any = false;
    uint32_t clamped = 0u;
    RawG raw;
    raw.g = (int32_t)gi;
    float4 q_raw = make_float4(0.f, 0.f, 0.f, 0.f);
    float v_opac_act = 0.f;
    const size_t cn = (size_t)a.C * a.N;
    for (uint32_t c = 0; c < a.C; ++c) {
        const size_t g = (size_t)c * a.N + gi;
        float Mo[15];
#pragma unroll
        for (int k = 0; k < 15; ++k) Mo[k] = 0.f;
        bool touched = false;
        auto load_raw = [&]() {   // issued before the records are read: these loads fly with the first records'
            if (!any) {
                raw.mu = {a.means[(size_t)gi * 3], a.means[(size_t)gi * 3 + 1], a.means[(size_t)gi * 3 + 2]};
                raw.q = reinterpret_cast<const float4*>(a.quats)[gi];
                raw.sc = {a.scales[(size_t)gi * 3], a.scales[(size_t)gi * 3 + 1], a.scales[(size_t)gi * 3 + 2]};
                any = true;
            }
            raw.opac = a.opacities[g];
        };
#define GSX_ADD_REC(M, R0, R1, R2, R3)                                                         \
        M[0] += R0.x; M[1] += R0.y; M[2] += R0.z; M[3] += R0.w;                                \
        M[4] += R1.x; M[5] += R1.y; M[6] += R1.z; M[7] += R1.w;                                \
        M[8] += R2.x; M[9] += R2.y; M[10] += R2.z; M[11] += R2.w;                              \
        M[12] += R3.x; M[13] += R3.y; M[14] += R3.z;
        if (ranges) {
            // ---- ranges: the records of this (camera, Gaussian) are the slots [first, cursor) ----
            constexpr int32_t RUN_T = 12;   // longer runs are summed by the whole wave
            const int32_t n_claimed = in ? ws_head[cn + g] : 0;   // plane 1: records the backward claimed
            const int32_t first = n_claimed > 0 ? range_first_slot(a, ws_head, g) : 0;
            // the same bound as the writers' guard (claim_record_slot's callers): a slot at or beyond rec_capacity was never written.  By
            // construction no run reaches it (a run is at most the Gaussian's rectangle of 16-px tiles, the slots are their sum); if the
            // invariant ever broke — a workspace packed for another frame — the gradient would lose records either way, but the gather must not
            // read unwritten or out-of-bounds memory on top of it
            const int32_t n = (int32_t)min((int64_t)n_claimed, max((int64_t)0, a.rec_capacity - (int64_t)first));
            const int32_t cursor = first + n;
            touched = n_claimed > 0;
            if (touched) {
                ws_head[cn + g] = 0;   // the run is consumed: the count is back at 0 for the next backward on this workspace
                load_raw();
            }
            if (n > 0 && n <= RUN_T) {
                for (int32_t r = first; r < cursor; r += 2) {   // two records in flight
                    const float4* p0 = ws_rec + (size_t)r * 4;
                    const bool two = r + 1 < cursor;
                    const float4* p1 = two ? p0 + 4 : p0;
                    const float4 a0 = p0[0], a1 = p0[1], a2 = p0[2], a3 = p0[3];
                    const float4 b0 = p1[0], b1 = p1[1], b2 = p1[2], b3 = p1[3];
                    GSX_ADD_REC(Mo, a0, a1, a2, a3)
                    if (two) { GSX_ADD_REC(Mo, b0, b1, b2, b3) }
                }
            }
            for (unsigned long long longs = __builtin_amdgcn_ballot_w64(n > RUN_T); longs != 0ull; longs &= longs - 1ull) {
                const int src = __builtin_ctzll(longs);
                const int32_t f = __builtin_amdgcn_readlane(first, src), nl = __builtin_amdgcn_readlane(n, src);
                float part[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) part[k] = 0.f;
                for (int32_t r = (int32_t)lane; r < nl; r += 64) {   // 64 consecutive records per round: 4 KB, coalesced
                    const float4* p0 = ws_rec + (size_t)(f + r) * 4;
                    const float4 a0 = p0[0], a1 = p0[1], a2 = p0[2], a3 = p0[3];
                    GSX_ADD_REC(part, a0, a1, a2, a3)
                }
                const float tot = butterfly_reduce16(part);   // lane 16 row + 4 quad holds the total of value 4 quad + {0,2,1,3}[row]
#pragma unroll
                for (int v = 0; v < 15; ++v) {
                    const int row = ((v & 3) == 1) ? 2 : (((v & 3) == 2) ? 1 : (v & 3));
                    const float t = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tot), row * 16 + (v >> 2) * 4));
                    Mo[v] += ((int)lane == src) ? t : 0.f;
                }
            }
        } else {
            int32_t it[NCH];
            int32_t all = -1;
#pragma unroll
            for (int k = 0; k < NCH; ++k) { it[k] = in ? ws_head[(size_t)k * cn + g] : -1; all &= it[k]; }
            touched = all >= 0;   // some chain has a record (an index has its sign bit clear)
            if (touched) {
#pragma unroll
                for (int k = 0; k < NCH; ++k)   // the chains are consumed: the head array is empty again for the next backward on this workspace
                    if (it[k] >= 0) ws_head[(size_t)k * cn + g] = -1;
                load_raw();
            }
            if (NCH == 1) {
                int32_t cur = it[0];
                while (cur >= 0) {
                    const float4* rec = ws_rec + (size_t)cur * 4;
                    const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3];
                    GSX_ADD_REC(Mo, r0, r1, r2, r3)
                    cur = __float_as_int(r3.w);
                }
            } else {
                // the chains side by side: every round issues the record loads of all chains some lane still walks before any is consumed
                // (a lane whose chain k has ended re-reads a record of one of its live chains — same line, no new traffic — and ignores it)
                while (all >= 0) {
                    int32_t live = it[0];
#pragma unroll
                    for (int k = 1; k < NCH; ++k) live = max(live, it[k]);
                    float4 r[NCH][4];
                    bool on[NCH];
#pragma unroll
                    for (int k = 0; k < NCH; ++k) {
                        on[k] = it[k] >= 0;
                        if (__builtin_amdgcn_ballot_w64(on[k]) != 0ull) {
                            const float4* rec = ws_rec + (size_t)(on[k] ? it[k] : live) * 4;
                            r[k][0] = rec[0]; r[k][1] = rec[1]; r[k][2] = rec[2]; r[k][3] = rec[3];
                        } else {
                            r[k][0] = r[k][1] = r[k][2] = r[k][3] = make_float4(0.f, 0.f, 0.f, 0.f);
                        }
                    }
                    all = -1;
#pragma unroll
                    for (int k = 0; k < NCH; ++k) {
                        if (on[k]) {
                            GSX_ADD_REC(Mo, r[k][0], r[k][1], r[k][2], r[k][3])
                            it[k] = __float_as_int(r[k][3].w);
                        }
                        all &= it[k];
                    }
                }
            }
        }
#undef GSX_ADD_REC
        if (!in) continue;
        if (!touched) {  // no tile touched this (camera, Gaussian): every output element is written, none needs a pre-fill
            v_colors[g * 3] = 0.f; v_colors[g * 3 + 1] = 0.f; v_colors[g * 3 + 2] = 0.f;
            if (!ACT) v_opacities[g] = 0.f;
            continue;
        }
        v_colors[g * 3] = Mo[0]; v_colors[g * 3 + 1] = Mo[1]; v_colors[g * 3 + 2] = Mo[2];
        if (ACT) v_opac_act = Mo[3] / raw.opac;
        else v_opacities[g] = Mo[3] / raw.opac;
        // moments -> (mean, quaternion, scale): once per (camera, Gaussian) (gsx_record.hpp: moments_to_gradients)
        const ShutterPoses sp(a.cams.viewmats0 + c * 16, nullptr);
        const CamFrame cf = make_cam_frame(sp);
        const float tb0[4] = {0.f, 0.f, 0.f, 0.f};
        FastRec r;
        {   // from here on raw.sc holds the conditioned scales (gsx_record.hpp: MAX_SCALE_RATIO; idempotent: a second camera adds no bits)
            const f3 sc_c = conditioned_scales(raw.sc);
            clamped |= clamped_axes(raw.sc, sc_c);
            raw.sc = sc_c;
        }
        make_record<true>(raw, cf, tb0, r);
        moments_to_gradients(raw, cf, r, Mo, geo, clamped);
    }
    if (!in) return;
    // One non-finite gradient poisons a Gaussian's Adam state for good, and a NaN opacity takes the next refine event's multinomial down with it (a device
    // assert).  The known source — records of splats thinner than fp32 can condition, gsx_record.hpp: MAX_SCALE_RATIO — is closed at the record; this is the
    // net under whatever else fp32 can do to ONE Gaussian: its geometric gradient is dropped for the iteration instead of written.
#pragma unroll
    for (int k = 0; k < 10; ++k) geo[k] = fabsf(geo[k]) < INFINITY ? geo[k] : 0.f;
    v_means[(size_t)gi * 3] = geo[0]; v_means[(size_t)gi * 3 + 1] = geo[1]; v_means[(size_t)gi * 3 + 2] = geo[2];
    if (!ACT) {
        reinterpret_cast<float4*>(v_quats)[gi] = make_float4(geo[3], geo[4], geo[5], geo[6]);
        v_scales[(size_t)gi * 3] = geo[7]; v_scales[(size_t)gi * 3 + 1] = geo[8]; v_scales[(size_t)gi * 3 + 2] = geo[9];
        return;
    }
    // ---- activation Jacobians (the arithmetic of splat_activations_bwd_kernel, gsx_sh.hip; an untouched Gaussian has zero gradients and
    // still gets its regulariser terms — it needs its activated scale / opacity for them: loaded here, the walk did not)
    if (!any || clamped != 0u) {   // (a clamped axis: the chain rule and the regulariser below want the TRUE activated scale back — a rare reload)
        raw.sc = {a.scales[(size_t)gi * 3], a.scales[(size_t)gi * 3 + 1], a.scales[(size_t)gi * 3 + 2]};
        if (!any) raw.opac = a.opacities[gi];
    }
    const float sc[3] = {raw.sc.x, raw.sc.y, raw.sc.z};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float gk = geo[7 + k] * sc[k];                                  // d exp(s) = exp(s)
        if (act.scale_reg != 0.f) gk = fmaf(act.scale_reg, sc[k], gk);
        act.v_scaling_raw[(size_t)gi * 3 + k] = gk;
    }
    q_raw = reinterpret_cast<const float4*>(act.rotation_raw)[gi];   // (loaded here, not in front of the record walk: four registers less across the walk and the chain rule keep the kernel at 4 waves / SIMD)
    const float nrm = sqrtf(q_raw.x * q_raw.x + q_raw.y * q_raw.y + q_raw.z * q_raw.z + q_raw.w * q_raw.w);
    float4 o;
    if (nrm > 1e-12f) {
        const float inv = 1.f / nrm;
        const float4 qn = make_float4(q_raw.x * inv, q_raw.y * inv, q_raw.z * inv, q_raw.w * inv);
        const float d = geo[3] * qn.x + geo[4] * qn.y + geo[5] * qn.z + geo[6] * qn.w;
        o = make_float4((geo[3] - d * qn.x) * inv, (geo[4] - d * qn.y) * inv, (geo[5] - d * qn.z) * inv, (geo[6] - d * qn.w) * inv);
    } else {
        o = make_float4(geo[3] * 1e12f, geo[4] * 1e12f, geo[5] * 1e12f, geo[6] * 1e12f);
    }
    reinterpret_cast<float4*>(act.v_rotation_raw)[gi] = o;
    const float sg = raw.opac, ds = sg * (1.f - sg);
    float go = v_opac_act * sg * (1.f - sg);
    if (act.opacity_reg != 0.f) go = fmaf(act.opacity_reg, ds, go);
    act.v_opacity_raw[gi] = go;
}

size_t raster_bwd_fast_workspace_bytes(uint32_t C, uint32_t N, int64_t n_isects) {
    // moment records (64 B per intersection) + chain heads (NSUB x 4 B per (camera, Gaussian)) + the forward's layout, 256 B aligned
    return align256((size_t)n_isects * 64) + head_planes_bytes(C, N) + raster_fwd_fast_workspace_bytes(C, N);
}

bool launch_raster_bwd_fast(int kind, RasterArgs a, const float* render_alphas, const int32_t* last_ids,
                            const float* v_render_colors, const float* v_render_alphas, float* v_means, float* v_quats,
                            float* v_scales, float* v_colors, float* v_opacities, void* workspace, size_t workspace_bytes,
                            const float4* packed_from_fwd, hipStream_t st, const uint8_t** tile_flags_out, const ActEpilogue* act) {
    *tile_flags_out = nullptr;
    if (workspace == nullptr || workspace_bytes < raster_bwd_fast_workspace_bytes(a.C, a.N, a.n_isects << (2 * a.lshift))) return false;
    const uint32_t n_tiles = a.tw * a.th;
    const dim3 grid(((n_tiles + 7u) / 8u) * 8u, a.C), block(RB);
    float4* ws_rec = (float4*)workspace;
    int32_t* ws_head = (int32_t*)((char*)workspace + align256(((size_t)a.n_isects << (2 * a.lshift)) * 64));   // lists per 32 x 32 pixels: four record slots per entry
    if (packed_from_fwd) {  // the forward of the same inputs left its packed records, tile flags and (empty) list heads with the caller
        a.packed = packed_from_fwd;
        if (kind == CAM_OPENCV_FISHEYE) a.tile_flags = ws_flags(packed_from_fwd, a.C, a.N);
        ws_head = raster_fwd_fast_heads(packed_from_fwd, a.C, a.N);   // all -1: set by the packer, restored by every gather
    } else {
        pack_into(kind, a, (char*)ws_head + head_planes_bytes(a.C, a.N), st, ws_head);  // also sets every chain head to -1 (and the mode word behind the planes to chains)
    }
    *tile_flags_out = a.tile_flags;
    a.rec_capacity = a.n_isects << (2 * a.lshift);
    a.rect_filter = 1u;   // (GSX_LIST_RECT=0 is a forward-only demonstration: a record run is sized by the Gaussian's tile rectangle)
    const dim3 ggrid((a.N + 255u) / 256u), gblock(256);
    // record chains per (camera, Gaussian): NSUB on frames of large footprints — lists per 32 x 32 pixels are chosen for exactly those, or more
    // than 8 tiles per Gaussian on average — else one (gsx_raster_common.hpp: tile_chain; GSX_BWD_CHAINS=1|4 forces it: tests, A/B)
    a.chain_mask = (a.lshift != 0u || a.n_isects_expected > 8 * (int64_t)a.C * (int64_t)a.N) ? (uint32_t)(NSUB - 1) : 0u;
    if (const char* e = test_switch("GSX_BWD_CHAINS")) a.chain_mask = atoi(e) > 1 ? (uint32_t)(NSUB - 1) : 0u;
    // Two backward kernels, same records and gather: Gaussian-major (the default; S-1M 0.49 vs 0.86 ms, S-5M @4K 1.9 vs 2.5 ms, garden
    // stand-in 248 vs 226 it/s) and pixel-major (GSX_BWD=pm forces it: tests, tools).
    const bool force_pm = [] { const char* e = test_switch("GSX_BWD"); return e && std::string(e) == "pm"; }();   // read per launch: the tests switch it
    const bool gaussian_major = !force_pm;
#define GSX_BLEND_BWD(KERNEL, KIND)                                                                                                        \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(KERNEL<KIND>), grid, block, 0, st, a, render_alphas, last_ids, v_render_colors, v_render_alphas, ws_rec, ws_head)
    if (gaussian_major) {
        if (kind == CAM_PERFECT_PINHOLE) GSX_BLEND_BWD(raster_bwd_gq_kernel, CAM_PERFECT_PINHOLE);
        else if (kind == CAM_OPENCV_PINHOLE) GSX_BLEND_BWD(raster_bwd_gq_kernel, CAM_OPENCV_PINHOLE);
        else GSX_BLEND_BWD(raster_bwd_gq_kernel, CAM_OPENCV_FISHEYE);
    } else {
        if (kind == CAM_PERFECT_PINHOLE) GSX_BLEND_BWD(raster_bwd_fast_kernel, CAM_PERFECT_PINHOLE);
        else if (kind == CAM_OPENCV_PINHOLE) GSX_BLEND_BWD(raster_bwd_fast_kernel, CAM_OPENCV_PINHOLE);
        else GSX_BLEND_BWD(raster_bwd_fast_kernel, CAM_OPENCV_FISHEYE);
    }
#undef GSX_BLEND_BWD
    // the moments -> gradient map only involves the camera pose: one instance serves every camera model
    // (act: the activation Jacobians as the gather's epilogue — the caller has checked C == 1 and that no reference-order kernel adds to the outputs afterwards)
    const ActEpilogue none{nullptr, nullptr, nullptr, nullptr, 0.f, 0.f};
#define GSX_GATHER(NCH, ACT) \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(gsx_bwd_gather_kernel<CAM_PERFECT_PINHOLE, NCH, ACT>), ggrid, gblock, 0, st, a, (const float4*)ws_rec, ws_head, v_means, v_quats, \
                       v_scales, v_colors, v_opacities, ACT ? *act : none)
    if (act != nullptr && act->rotation_raw != nullptr) {
        if (a.chain_mask) GSX_GATHER(NSUB, true);
        else GSX_GATHER(1, true);
    } else {
        if (a.chain_mask) GSX_GATHER(NSUB, false);
        else GSX_GATHER(1, false);
    }
#undef GSX_GATHER
    return true;
}

}  // namespace gsx

#ifdef GSX_STATS
extern "C" void gsx_debug_read_stats(unsigned long long* out, int reset) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(gsx::g_stats), sizeof(unsigned long long) * 16);
    if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(gsx::g_stats), z, sizeof(z)); }
}
#endif
#ifdef GSX_CLOCKS
extern "C" void gsx_debug_read_clocks(unsigned long long* out, int reset) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(gsx::g_clk), sizeof(unsigned long long) * 12);
    if (reset) { unsigned long long z[12] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(gsx::g_clk), z, sizeof(z)); }
}
#endif
