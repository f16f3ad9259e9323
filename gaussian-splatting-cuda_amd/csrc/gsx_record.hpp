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
    float Rc[3][3];   // camera -> world "rotation" (== reference's R_inv = mat3_cast(inverse(quat_cast(R))), Cameras.cuh:262) — see Rci
    float Rci[3][3];  // its EXACT inverse (world -> camera), not its transpose
    f3 c;             // camera centre in world space
};

// Round 6.  The reference takes a pose through fp32 quaternions (matrix -> quat_cast -> inverse -> mat3_cast, Cameras.cuh:42-52,258-262): the
// R_inv every ray is built from is orthonormal only to fp32 rounding — 7e-7 for the diagonal S-8cam cameras (3 / 5: the y-branch of
// quat_cast, |q|^2 - 1 = 1.2e-7), 0 for an axis-aligned pose.  The reference never needs the inverse of R_inv (it intersects world-space
// rays: origin o = -R_inv t, direction R_inv p); the Delta-form does — the Gaussian's camera-space centre m with R_inv m = mu - o —, and
// rounds 1 - 5 took m = R_inv^T (mu - o): with a 7e-7 non-orthonormal R_inv every Gaussian sat 7e-7 |mu - o| = 7e-6 world units from where the
// reference's rays see it — 3.5e-3 sigma of the smallest S-1M Gaussians, direction dependent: on cameras 3 / 5 HIP was 1.3 - 1.6e-3 (gradient
// rel-L2) and 10 000 alpha pixels > 1e-4 from the reference kernel, on the axis-aligned cfg2 camera 4e-4 / 28 (profiles/parity_r06.md).
// m = R_inv^-1 (mu - o) with the inverse formed in double (adjugate / determinant, once per thread: ~60 DP operations) removes it.
GSX_DEV CamFrame make_cam_frame(const ShutterPoses& sp) {
    CamFrame f;
    const m33 Rinv = quat_to_mat_raw(quat_conj_over_norm2(sp.q0));
    for (int r = 0; r < 3; ++r)
        for (int k = 0; k < 3; ++k) f.Rc[r][k] = Rinv.a[r][k];
    const f3 rt = mul(Rinv, sp.t0);
    f.c = {-rt.x, -rt.y, -rt.z};
#ifdef GSX_FRAME_TRANSPOSE   // A/B only (tools/): rounds 1 - 5's R_inv^T
    for (int r = 0; r < 3; ++r)
        for (int k = 0; k < 3; ++k) f.Rci[r][k] = Rinv.a[k][r];
    return f;
#endif
    const double a00 = Rinv.a[0][0], a01 = Rinv.a[0][1], a02 = Rinv.a[0][2], a10 = Rinv.a[1][0], a11 = Rinv.a[1][1], a12 = Rinv.a[1][2],
                 a20 = Rinv.a[2][0], a21 = Rinv.a[2][1], a22 = Rinv.a[2][2];
    const double c00 = a11 * a22 - a12 * a21, c01 = a12 * a20 - a10 * a22, c02 = a10 * a21 - a11 * a20;
    const double idet = 1.0 / (a00 * c00 + a01 * c01 + a02 * c02);
    f.Rci[0][0] = (float)(c00 * idet); f.Rci[0][1] = (float)((a02 * a21 - a01 * a22) * idet); f.Rci[0][2] = (float)((a01 * a12 - a02 * a11) * idet);
    f.Rci[1][0] = (float)(c01 * idet); f.Rci[1][1] = (float)((a00 * a22 - a02 * a20) * idet); f.Rci[1][2] = (float)((a02 * a10 - a00 * a12) * idet);
    f.Rci[2][0] = (float)(c02 * idet); f.Rci[2][1] = (float)((a01 * a20 - a00 * a21) * idet); f.Rci[2][2] = (float)((a00 * a11 - a01 * a10) * idet);
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
    f3 a0, a1, a2, h, B0, B1, c01, c12, c20, m, q0, q1;   // (q0, q1): the orthonormal pair of span(B0, B1) the factor L was taken against (B = Q L)
    float inv_d0, sL;
    float Mt[3][3];  // M(r,c) = (1/s_r) R(c,r)
};

// Round 6: the Delta-form's conditioning limit.  Its record is built from the cofactors of A = diag(1/s) R^T Rc, whose entries differ by the SQUARE of
// the scale ratio: up to ~1e4 : 1 (needles, discs a trained model normally holds) fp32 carries it — the regimes of tests/test_gpu_reference_hip.py —,
// at 3e4 : 1 the forward is 8e-3 off and the Gaussian-major backward returns NaN, at 3e5 : 1 the image is 0.05 off (tools/pancake_probe.py).  Long MCMC
// runs DO get there: the scale regulariser drives the normal of a flat surface splat to 1e-6 .. 1e-9 of its extent (25 000 iterations of
// examples/train_synthetic.py: ratios to 2e7, and every second run ended in non-finite parameters).  A disc of thickness s_max / 8192 and one of
// thickness 0 are the same splat to a pixel's ray — the response is the in-plane Gaussian at the piercing point, the thickness enters with
// (s_thin / s_wide)^2 tan^2(incidence) ~ 1.5e-8 tan^2 —, so the record is built from scales clamped to s_max / 8192 from below: a no-op (bit for bit)
// for every ratio below it, the reference's image (1.6e-4, the threshold-decision floor) beyond it.  The clamped axis carries no scale gradient.
constexpr float MAX_SCALE_RATIO = 8192.f;
GSX_DEV f3 conditioned_scales(f3 s) {
    const float lo = fmaxf(fmaxf(s.x, s.y), s.z) * (1.f / MAX_SCALE_RATIO);
    return {fmaxf(s.x, lo), fmaxf(s.y, lo), fmaxf(s.z, lo)};
}

// bit k: axis k of `s` is held by the clamp (its scale gradient is dropped)
GSX_DEV uint32_t clamped_axes(f3 s, f3 c) { return (s.x < c.x ? 1u : 0u) | (s.y < c.y ? 2u : 0u) | (s.z < c.z ? 4u : 0u); }

// r.sc: CONDITIONED scales (the callers — store_packed_record, the backward's gather — pass conditioned_scales(scales))
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
    const float mx = cf.Rci[0][0] * dm.x + cf.Rci[0][1] * dm.y + cf.Rci[0][2] * dm.z;   // m = R_inv^-1 (mu - o): make_cam_frame
    const float my = cf.Rci[1][0] * dm.x + cf.Rci[1][1] * dm.y + cf.Rci[1][2] * dm.z;
    const float mz = cf.Rci[2][0] * dm.x + cf.Rci[2][1] * dm.y + cf.Rci[2][2] * dm.z;
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
        o.m = {mx, my, mz}; o.inv_d0 = inv_d0; o.sL = sL;
        // guarded like i00 / i11 of moments_to_gradients: B0 = 0 or B1 parallel to B0 in fp32 (a collapsed scale) must give a zero
        // direction, not 0 * inf = NaN — one NaN gradient would poison the Gaussian's Adam state for good
        const float nr = sqrtf(dot3(rr, rr));
        const f3 zero3{0.f, 0.f, 0.f};
        o.q0 = n0 > 0.f ? B0 * (1.f / n0) : zero3; o.q1 = (nr > 0.f && nr < INFINITY) ? rr * (1.f / nr) : zero3;   // (selects: a NaN rr must not pass through a multiply)
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int k = 0; k < 3; ++k) o.Mt[i][k] = M[i][k];
    }
}

// ---- the backward's chain rule: moments -> (mean, quaternion, scale) -------------------------------------------------------------------
// The blend backward hands over, per (camera, Gaussian), 11 geometric moments of its pixel weights in the WHITENED pixel offsets
//     x0 = l00 du + l01 dv,   x1 = l11 dv          (what the alpha evaluation computes anyway: N ~ x0^2 + x1^2)
//     Wa = sum a {x0^2, x0 x1, x1^2, x0, x1},      Wb = sum b {1, x0, x1, x0^2, x0 x1, x1^2}          (a = -2 (dL/dD) / den', b = a D 0.5 log2 e)
// and this function maps them to the gradients.  Rounds 1 - 4 took the moments in (du, dv) and went through the cofactors of A; round 5 found
// what that costs on NEEDLES (scale ratios beyond ~100 : 1: a regime trained models reach and random scenes do not): a thin footprint makes
// sum a (du, dv)(du, dv)^T nearly rank one, the gradient of the short axes lives in its SMALL eigenvalue, and the cofactor columns differ by the
// square of the scale ratio — rounding the moments to fp32 alone cost 2e-2 of the short axes' gradients at 600 : 1, the map itself 5e-2.
// The form used now has no cancellation to lose digits in:
//   * with Gamma = dA A^-1 (A = diag(1/s) R^T Rc: d(scale k) is Gamma = -(ds_k / s_k) e_k e_k^T, a rotation is S^-1 [w]x S),
//         dN = 2 tr(Gamma) N - 2 V^T Gamma V        (V = (A p) x (A m) = cof(A) (p x m), cof((1 + Gamma) A) = (1 + tr Gamma - Gamma^T) cof(A))
//         dDn = 2 y^T Gamma y                       (y = A p)
//     so  G_Gamma = sum 2 vN (N 1 - V V^T) + sum 2 vD y y^T,   v_scale[k] = -G_Gamma[k][k] / s_k,   G_M = G_Gamma S R^T;
//   * V = Q x with Q the orthonormal pair the triangular factor was taken against (B = Q L), y = [h, a~0, a~1] (1, x0, x1) with
//     [a~0 a~1] = [a0 a1] L^-1:   sum 2 vN V V^T = Q Wa Q^T,  sum 2 vN N = tr Wa,  sum 2 vD y y^T = Y~ Wb Y~^T  — sums of products of well-scaled factors;
//   * the mean sees N only (Dn = |A p|^2 does not depend on it): through (u0, v0) in the offsets and through B = mz cof(A) [beta0 beta1].
// Checked against torch autograd in float64 to 1e-12; in fp32 on 600 : 1 needles the scale gradients are within 4e-4 of float64 and 100 : 1 within
// 5e-5 — what is left is the fp32 record the PIXELS were evaluated with, a double-precision chain gives the same numbers
// (tests/test_gpu_reference_hip.py::test_trained_model_regimes_vs_reference[needles]: the tensor's rel-L2 against the reference kernel 3.2e-3 -> below 1e-4).
// Mo[4..14] = (Wa00, Wa01, Wa11, Wa0, Wa1, Wb, Wb0, Wb1, Wb00, Wb01, Wb11); fisheye: with the pixel's w folded in as the kernels do (the map is
// the same).  geo[0..2] += v_mean, geo[3..6] += v_quat (raw, wxyz), geo[7..9] += v_scale.
// raw.sc: the conditioned scales the record was made from; `clamped` = clamped_axes(): those axes get no scale gradient.
GSX_DEV void moments_to_gradients(const RawG& raw, const CamFrame& cf, const FastRec& r, const float* __restrict__ Mo, float* __restrict__ geo, uint32_t clamped = 0u) {
    const float kap = -r.inv_d0;                 // 2 / d0 times the -1/2 of dalpha/dD that the blend kernel leaves out of its weights
    const float kb = kap / HALF_LOG2E;           // b was accumulated with D scaled by 0.5 log2 e
    const float isL = 1.f / r.sL;                // the kernels' x carries sL = sqrt(0.5 log2 e / d0)
    const float i00 = r.l00 > 0.f ? 1.f / r.l00 : 0.f, i11 = r.l11 > 0.f ? 1.f / r.l11 : 0.f, i01 = -r.l01 * i00 * i11;   // L^-1: du = i00 x0 + i01 x1, dv = i11 x1
    const f3 q0 = r.q0, q1 = r.q1;
    // ---- G_Gamma
    const float cN = kap * isL * isL;
    const float Wa00 = Mo[4] * cN, Wa01 = Mo[5] * cN, Wa11 = Mo[6] * cN, trW = Wa00 + Wa11;
    const f3 qa = q0 * Wa00 + q1 * Wa01, qb = q0 * Wa01 + q1 * Wa11;        // Q Wa
    const f3 at0 = r.a0 * i00, at1 = r.a0 * i01 + r.a1 * i11;               // dy / dx0, dy / dx1
    const float Wb = Mo[9] * -kb, Wb0 = Mo[10] * -kb, Wb1 = Mo[11] * -kb, Wb00 = Mo[12] * -kb, Wb01 = Mo[13] * -kb, Wb11 = Mo[14] * -kb;
    const f3 yb0 = r.h * Wb + at0 * Wb0 + at1 * Wb1, yb1 = r.h * Wb0 + at0 * Wb00 + at1 * Wb01, yb2 = r.h * Wb1 + at0 * Wb01 + at1 * Wb11;   // Y~ Wb
    const float q0v[3] = {q0.x, q0.y, q0.z}, q1v[3] = {q1.x, q1.y, q1.z}, qav[3] = {qa.x, qa.y, qa.z}, qbv[3] = {qb.x, qb.y, qb.z};
    const float hv[3] = {r.h.x, r.h.y, r.h.z}, t0v[3] = {at0.x, at0.y, at0.z}, t1v[3] = {at1.x, at1.y, at1.z};
    const float y0v[3] = {yb0.x, yb0.y, yb0.z}, y1v[3] = {yb1.x, yb1.y, yb1.z}, y2v[3] = {yb2.x, yb2.y, yb2.z};
    float G[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            G[i][j] = (i == j ? trW : 0.f) - (qav[i] * q0v[j] + qbv[i] * q1v[j]) + (y0v[i] * hv[j] + y1v[i] * t0v[j] + y2v[i] * t1v[j]);
    // ---- scale and rotation: quat_scale_to_preci_half_vjp (Utils.cuh:104-158) with v_M = vMt^T, vMt = G_Gamma S R^T (M here is the reference's Mt)
    const float sv[3] = {raw.sc.x, raw.sc.y, raw.sc.z};
    const float isv[3] = {1.f / raw.sc.x, 1.f / raw.sc.y, 1.f / raw.sc.z};
#pragma unroll
    for (int k = 0; k < 3; ++k) geo[7 + k] += ((clamped >> k) & 1u) ? 0.f : -G[k][k] * isv[k];
    float vMt[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k)   // s_j R(k,j) = s_j^2 Mt[j][k]
            vMt[i][k] = G[i][0] * (sv[0] * sv[0] * r.Mt[0][k]) + G[i][1] * (sv[1] * sv[1] * r.Mt[1][k]) + G[i][2] * (sv[2] * sv[2] * r.Mt[2][k]);
    float w = raw.q.x, x_ = raw.q.y, y_ = raw.q.z, z_ = raw.q.w;
    const float inv_norm = rsqrtf(x_ * x_ + y_ * y_ + z_ * z_ + w * w);
    w *= inv_norm; x_ *= inv_norm; y_ *= inv_norm; z_ *= inv_norm;
#define GSX_G(i, j) (vMt[i][j] * isv[i])
    float vq[4];
    vq[0] = 2.f * (x_ * (GSX_G(1, 2) - GSX_G(2, 1)) + y_ * (GSX_G(2, 0) - GSX_G(0, 2)) + z_ * (GSX_G(0, 1) - GSX_G(1, 0)));
    vq[1] = 2.f * (-2.f * x_ * (GSX_G(1, 1) + GSX_G(2, 2)) + y_ * (GSX_G(0, 1) + GSX_G(1, 0)) + z_ * (GSX_G(0, 2) + GSX_G(2, 0)) + w * (GSX_G(1, 2) - GSX_G(2, 1)));
    vq[2] = 2.f * (x_ * (GSX_G(0, 1) + GSX_G(1, 0)) - 2.f * y_ * (GSX_G(0, 0) + GSX_G(2, 2)) + z_ * (GSX_G(1, 2) + GSX_G(2, 1)) + w * (GSX_G(2, 0) - GSX_G(0, 2)));
    vq[3] = 2.f * (x_ * (GSX_G(0, 2) + GSX_G(2, 0)) + y_ * (GSX_G(1, 2) + GSX_G(2, 1)) - 2.f * z_ * (GSX_G(0, 0) + GSX_G(1, 1)) + w * (GSX_G(0, 1) - GSX_G(1, 0)));
#undef GSX_G
    const float qn[4] = {w, x_, y_, z_};
    const float dq = vq[0] * qn[0] + vq[1] * qn[1] + vq[2] * qn[2] + vq[3] * qn[3];
#pragma unroll
    for (int k = 0; k < 4; ++k) geo[3 + k] += (vq[k] - dq * qn[k]) * inv_norm;
    // ---- mean: N = |du B0 + dv B1|^2 with (du, dv) = (u, v) - (u0, v0), B0 = mz (c20 - v0 c01), B1 = mz (u0 c01 - c12), (u0, v0) = (mx, my) / mz
    const float kf = kap * isL;
    const f3 t = (q0 * Mo[7] + q1 * Mo[8]) * kf;                                                 // sum 2 vN V
    const f3 G_B0 = q0 * (kf * (Mo[4] * i00 + Mo[5] * i01)) + q1 * (kf * (Mo[5] * i00 + Mo[6] * i01));   // sum 2 vN V du
    const f3 G_B1 = (q0 * Mo[5] + q1 * Mo[6]) * (kf * i11);                                      // sum 2 vN V dv
    const float mz = r.m.z, imz = 1.f / mz;
    const float G_u0 = -dot3(r.B0, t) + mz * dot3(r.c01, G_B1);
    const float G_v0 = -dot3(r.B1, t) - mz * dot3(r.c01, G_B0);
    const float G_mx = G_u0 * imz, G_my = G_v0 * imz;
    const float G_mz = (trW - (r.u0 * G_u0 + r.v0 * G_v0)) * imz;                               // sum 2 vN N / mz through B, then through (u0, v0)
    // m = Rci (mu - c)  ->  v_mean = Rci^T G_m
    geo[0] += cf.Rci[0][0] * G_mx + cf.Rci[1][0] * G_my + cf.Rci[2][0] * G_mz;
    geo[1] += cf.Rci[0][1] * G_mx + cf.Rci[1][1] * G_my + cf.Rci[2][1] * G_mz;
    geo[2] += cf.Rci[0][2] * G_mx + cf.Rci[1][2] * G_my + cf.Rci[2][2] * G_mz;
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
    RawG rc = raw;
    rc.sc = conditioned_scales(raw.sc);
    make_record<false>(rc, cf, tb0, r);
    if (!(r.hx > -INFINITY) && !(r.lo + LOG2_255 > 0.f)) r.lo = -INFINITY;  // never visible (opacity <= 1/255)
    if (!(fabsf(r.l00) < INFINITY)) r.lo = -INFINITY;                       // camera-space z == 0: skipped
    o[0] = make_float4(r.u0, r.v0, r.l00, r.l01);
    o[1] = make_float4(r.l11, r.lo, r.d1, r.d2);
    o[2] = make_float4(r.d3, r.d4, r.d5, raw.rgb.x);
    o[3] = make_float4(raw.rgb.y, raw.rgb.z, __uint_as_float(rect_x), __uint_as_float(rect_y));
}

}  // namespace gsx
