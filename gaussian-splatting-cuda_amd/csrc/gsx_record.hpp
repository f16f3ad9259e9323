// gsx_record.hpp — the packed 64 B camera-space record of one (camera, Gaussian) pair: what the fast blend kernels stage per tile
// intersection (gsx_raster_fast.hip) and what the fused front end writes while it has the Gaussian in registers (gsx_frontend.hip).
//   p0 = (u0, v0, l00, l01)  p1 = (l11, lo, d1, d2)  p2 = (d3, d4, d5, red)  p3 = (green, blue, -, -)
// Algebra: header comment of gsx_raster_fast.hip (item 1).
#pragma once
#include "gsx_device.hpp"

namespace gsx {

constexpr float LOG2_255 = 7.994353436858858f;
constexpr float HALF_LOG2E = 0.7213475204444817f;  // 0.5 * log2(e)

struct CamFrame {
    float Rc[3][3];  // camera -> world rotation (== reference's R_inv, Cameras.cuh:262)
    f3 c;            // camera centre in world space
};

GSX_DEV CamFrame make_cam_frame(const ShutterPoses& sp) {
    CamFrame f;
    const m33 Rinv = quat_to_mat_raw(quat_conj_over_norm2(sp.q0));
    for (int r = 0; r < 3; ++r)
        for (int k = 0; k < 3; ++k) f.Rc[r][k] = Rinv.a[r][k];
    const f3 rt = mul(Rinv, sp.t0);
    f.c = {-rt.x, -rt.y, -rt.z};
    return f;
}

// raw per-Gaussian parameters (inputs of make_record: pack kernel and the backward's gather kernel)
struct RawG {
    f3 mu; float4 q; f3 sc; float opac; f3 rgb; int32_t g;
};

// Everything the pixel loop needs for one Gaussian (see header comment).  tb = tile bounds in (u,v).
struct FastRec {
    float u0, v0, hx, hy;        // footprint centre and conservative half extents in (u,v)
    float l00, l01, l11, lo;     // triangular factor (pre-scaled by sqrt(0.5 log2 e / d0)), log2(opacity)
    float d1, d2, d3, d4, d5;    // |A p|^2 / |A p_mu|^2 = 1 + d1 du + d2 dv + d3 du^2 + d4 du dv + d5 dv^2
    // backward finishing only: columns of A, h = A p_mu, B0, B1, the cofactor columns, camera-space centre, 1/d0
    f3 a0, a1, a2, h, B0, B1, c01, c12, c20, m;
    float inv_d0;
    float Mt[3][3];  // M(r,c) = (1/s_r) R(c,r)
};

template <bool BWD>
GSX_DEV void make_record(const RawG& r, const CamFrame& cf, const float tb[4], FastRec& o) {
    const m33 R = quat_to_rotmat(r.q.x, r.q.y, r.q.z, r.q.w);
    const float is[3] = {1.f / r.sc.x, 1.f / r.sc.y, 1.f / r.sc.z};
    float M[3][3], A[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) M[i][k] = is[i] * R.a[k][i];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) A[i][j] = M[i][0] * cf.Rc[0][j] + M[i][1] * cf.Rc[1][j] + M[i][2] * cf.Rc[2][j];
    const f3 dm = r.mu - cf.c;
    const float mx = cf.Rc[0][0] * dm.x + cf.Rc[1][0] * dm.y + cf.Rc[2][0] * dm.z;
    const float my = cf.Rc[0][1] * dm.x + cf.Rc[1][1] * dm.y + cf.Rc[2][1] * dm.z;
    const float mz = cf.Rc[0][2] * dm.x + cf.Rc[1][2] * dm.y + cf.Rc[2][2] * dm.z;
    const float imz = 1.f / mz;
    o.u0 = mx * imz; o.v0 = my * imz;
    const f3 a0{A[0][0], A[1][0], A[2][0]}, a1{A[0][1], A[1][1], A[2][1]}, a2{A[0][2], A[1][2], A[2][2]};
    const f3 c01 = cross3(a0, a1), c12 = cross3(a1, a2), c20 = cross3(a2, a0);
    const f3 B0 = (c20 - c01 * o.v0) * mz;
    const f3 B1 = (c01 * o.u0 - c12) * mz;
    const f3 h = a0 * o.u0 + a1 * o.v0 + a2;
    const float d0 = dot3(h, h);
    const float inv_d0 = 1.f / d0;
    o.d1 = 2.f * dot3(h, a0) * inv_d0; o.d2 = 2.f * dot3(h, a1) * inv_d0;
    o.d3 = dot3(a0, a0) * inv_d0; o.d4 = 2.f * dot3(a0, a1) * inv_d0; o.d5 = dot3(a1, a1) * inv_d0;
    const float sL = sqrtf(HALF_LOG2E * inv_d0);
    const float n0 = sqrtf(dot3(B0, B0));
    const float l01r = dot3(B0, B1) / n0;
    const f3 rr = B1 - B0 * (l01r / n0);
    o.l00 = n0 * sL; o.l01 = l01r * sL; o.l11 = sqrtf(dot3(rr, rr)) * sL;
    o.lo = __log2f(r.opac);
    // conservative footprint: |L d|^2 <= tau2 * max_tile den'
    const float tau2 = o.lo + LOG2_255;
    float hx = -INFINITY, hy = -INFINITY;
    if (tau2 > 0.f && fabsf(mz) > 1e-12f) {
        float dmax = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float du = ((k & 1) ? tb[1] : tb[0]) - o.u0, dv = ((k & 2) ? tb[3] : tb[2]) - o.v0;
            dmax = fmaxf(dmax, 1.f + du * (o.d1 + o.d3 * du + o.d4 * dv) + dv * (o.d2 + o.d5 * dv));
        }
        const float rad = sqrtf(tau2 * dmax) * 1.001f + 1e-7f;
        hy = rad / o.l11;
        hx = rad * sqrtf(o.l01 * o.l01 + o.l11 * o.l11) / (o.l00 * o.l11);
        if (!(hx == hx) || !(hy == hy)) { hx = INFINITY; hy = INFINITY; }  // degenerate factor: never cull
    }
    o.hx = hx; o.hy = hy;
    if (BWD) {
        o.a0 = a0; o.a1 = a1; o.a2 = a2; o.h = h; o.B0 = B0; o.B1 = B1; o.c01 = c01; o.c12 = c12; o.c20 = c20;
        o.m = {mx, my, mz}; o.inv_d0 = inv_d0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int k = 0; k < 3; ++k) o.Mt[i][k] = M[i][k];
    }
}

// Writes the packed record of one (camera, Gaussian) (see the layout above).  `lo` = -inf marks a Gaussian no pixel can see (opacity
// <= 1/255) or whose camera-space z is exactly 0 (skipped: DESIGN.md §8).
// The two spare floats of the record carry the Gaussian's rectangle of 16-pixel tiles [x0, x1) x [y0, y1) as the intersection computes
// it from means2d / radii (IntersectTile.cu:65-76), 16 bits per bound: rect_x = x0 | x1 << 16, rect_y = y0 | y1 << 16.  Only read when
// the LISTS were built per 32 x 32 pixels (RasterArgs::lshift): a 16-pixel tile then skips the entries of its parent's list whose
// rectangle does not contain it, so that exactly the reference's (tile, Gaussian) pairs are composited.  RECT_ALL = no restriction
// (records packed without the projection's outputs: pack_records_kernel).
constexpr uint32_t RECT_ALL = 0xFFFF0000u;
GSX_DEV uint32_t tile16_range(float m, float r, uint32_t n_tiles) {   // the arithmetic of tile_rect (gsx_intersect.hip), tile size 16
    const float t = m / 16.f, tr = r / 16.f;
    const float lo = floorf(t - tr), hi = ceilf(t + tr);
    const uint32_t a = lo > 0.f ? (lo >= 65535.f ? 65535u : (uint32_t)lo) : 0u, b = hi > 0.f ? (hi >= 65535.f ? 65535u : (uint32_t)hi) : 0u;
    return min(a, n_tiles) | (min(b, n_tiles) << 16);
}
GSX_DEV bool rect_has_tile(float4 r3, uint32_t tile_x, uint32_t tile_y) {
    const uint32_t rx = __float_as_uint(r3.z), ry = __float_as_uint(r3.w);
    return tile_x >= (rx & 0xFFFFu) && tile_x < (rx >> 16) && tile_y >= (ry & 0xFFFFu) && tile_y < (ry >> 16);
}

// A record no pixel can see (lo = -inf: alpha = 0 everywhere, empty footprint, empty rectangle): what the fused front end writes for a
// Gaussian its projection culled, so that a workspace never holds uninitialised records.
GSX_DEV void store_null_record(float4* __restrict__ o) {
    o[0] = make_float4(0.f, 0.f, 1.f, 0.f);
    o[1] = make_float4(1.f, -INFINITY, 0.f, 0.f);
    o[2] = make_float4(0.f, 0.f, 0.f, 0.f);
    o[3] = make_float4(0.f, 0.f, 0.f, 0.f);
}

GSX_DEV void store_packed_record(const RawG& raw, const CamFrame& cf, float4* __restrict__ o, uint32_t rect_x = RECT_ALL, uint32_t rect_y = RECT_ALL) {
    const float tb0[4] = {0.f, 0.f, 0.f, 0.f};
    FastRec r;
    make_record<false>(raw, cf, tb0, r);
    if (!(r.hx > -INFINITY) && !(r.lo + LOG2_255 > 0.f)) r.lo = -INFINITY;  // never visible (opacity <= 1/255)
    if (!(fabsf(r.l00) < INFINITY)) r.lo = -INFINITY;                       // camera-space z == 0: skipped
    o[0] = make_float4(r.u0, r.v0, r.l00, r.l01);
    o[1] = make_float4(r.l11, r.lo, r.d1, r.d2);
    o[2] = make_float4(r.d3, r.d4, r.d5, raw.rgb.x);
    o[3] = make_float4(raw.rgb.y, raw.rgb.z, __uint_as_float(rect_x), __uint_as_float(rect_y));
}

}  // namespace gsx
