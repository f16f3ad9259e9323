// gsx_device.hpp — device-side math shared by the gfx950 kernels: small vector helpers, the
// quaternion / rotation conventions of the reference, and the three camera models.
//
// What it restates (reference paths relative to /root/reference):
//   quaternion pose from a row-major viewmat ........ gsplat/Cameras.cuh:33-71
//   shutter pose interpolation ....................... gsplat/Cameras.cuh:268-280
//   perfect pinhole / OpenCV pinhole / OpenCV fisheye  gsplat/Cameras.cuh:416-471, 473-755, 817-1001
//   quat_to_rotmat (wxyz, renormalising) ............. gsplat/Utils.cuh:80-102
// glm is not used anywhere: the few glm semantics that carry numerics are written out
// (SURVEY.md Appendix B).  Matrices are plain row-major float[3][3] ("math" indexing).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gsx.h"

namespace gsx {

// 16 B nontemporal (streaming) global accesses: for data a kernel touches exactly once (optimizer state, one-shot records)
typedef float gsx_nt4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float4* p) {
    const gsx_nt4 t = __builtin_nontemporal_load(reinterpret_cast<const gsx_nt4*>(p));
    return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void nt_store4(float4 v, float4* p) { __builtin_nontemporal_store(gsx_nt4{v.x, v.y, v.z, v.w}, reinterpret_cast<gsx_nt4*>(p)); }

#define GSX_DEV __device__ __forceinline__

struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct quat { float w, x, y, z; };
struct m33 { float a[3][3]; };

GSX_DEV f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
GSX_DEV f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
GSX_DEV f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
GSX_DEV f3 operator*(float s, f3 a) { return {a.x * s, a.y * s, a.z * s}; }
GSX_DEV float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
GSX_DEV f3 cross3(f3 a, f3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
GSX_DEV f3 mul(const m33& m, f3 v) {
    return {m.a[0][0] * v.x + m.a[0][1] * v.y + m.a[0][2] * v.z,
            m.a[1][0] * v.x + m.a[1][1] * v.y + m.a[1][2] * v.z,
            m.a[2][0] * v.x + m.a[2][1] * v.y + m.a[2][2] * v.z};
}

// rotation matrix of a quaternion WITHOUT normalising it (== glm::mat3_cast)
GSX_DEV m33 quat_to_mat_raw(quat q) {
    const float xx = q.x * q.x, yy = q.y * q.y, zz = q.z * q.z;
    const float xz = q.x * q.z, xy = q.x * q.y, yz = q.y * q.z;
    const float wx = q.w * q.x, wy = q.w * q.y, wz = q.w * q.z;
    m33 r;
    r.a[0][0] = 1.f - 2.f * (yy + zz); r.a[0][1] = 2.f * (xy - wz);       r.a[0][2] = 2.f * (xz + wy);
    r.a[1][0] = 2.f * (xy + wz);       r.a[1][1] = 1.f - 2.f * (xx + zz); r.a[1][2] = 2.f * (yz - wx);
    r.a[2][0] = 2.f * (xz - wy);       r.a[2][1] = 2.f * (yz + wx);       r.a[2][2] = 1.f - 2.f * (xx + yy);
    return r;
}
// gsplat/Utils.cuh:80-102: normalise (rsqrt) then convert
GSX_DEV m33 quat_to_rotmat(float w, float x, float y, float z) {
    const float inv = rsqrtf(x * x + y * y + z * z + w * w);
    return quat_to_mat_raw(quat{w * inv, x * inv, y * inv, z * inv});
}
GSX_DEV quat quat_conj_over_norm2(quat q) {  // glm::inverse(quat)
    const float d = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
    return {q.w / d, -q.x / d, -q.y / d, -q.z / d};
}
GSX_DEV f3 quat_rotate(quat q, f3 v) {  // glm::rotate(quat, vec3)
    const f3 u{q.x, q.y, q.z};
    const f3 uv = cross3(u, v);
    const f3 uuv = cross3(u, uv);
    return v + ((uv * q.w) + uuv) * 2.f;
}
// rotation part of a row-major [4,4] world->camera matrix -> quaternion (largest-component branch)
GSX_DEV quat quat_from_viewmat(const float* __restrict__ se3) {
    const float m00 = se3[0], m01 = se3[1], m02 = se3[2];
    const float m10 = se3[4], m11 = se3[5], m12 = se3[6];
    const float m20 = se3[8], m21 = se3[9], m22 = se3[10];
    const float fx = m00 - m11 - m22, fy = m11 - m00 - m22, fz = m22 - m00 - m11, fw = m00 + m11 + m22;
    int bi = 0;
    float big = fw;
    if (fx > big) { big = fx; bi = 1; }
    if (fy > big) { big = fy; bi = 2; }
    if (fz > big) { big = fz; bi = 3; }
    const float bv = sqrtf(big + 1.f) * 0.5f;
    const float mult = 0.25f / bv;
    switch (bi) {
    case 0: return {bv, (m21 - m12) * mult, (m02 - m20) * mult, (m10 - m01) * mult};
    case 1: return {(m21 - m12) * mult, bv, (m10 + m01) * mult, (m02 + m20) * mult};
    case 2: return {(m02 - m20) * mult, (m10 + m01) * mult, bv, (m21 + m12) * mult};
    default: return {(m10 - m01) * mult, (m02 + m20) * mult, (m21 + m12) * mult, bv};
    }
}
GSX_DEV quat quat_slerp(quat a, quat b, float t) {  // glm::slerp
    float c = a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    if (c < 0.f) { b = {-b.w, -b.x, -b.y, -b.z}; c = -c; }
    if (c > 1.f - 1.1920929e-7f) {
        const float s = 1.f - t;
        return {a.w * s + b.w * t, a.x * s + b.x * t, a.y * s + b.y * t, a.z * s + b.z * t};
    }
    const float ang = acosf(c);
    const float s0 = sinf((1.f - t) * ang), s1 = sinf(t * ang), sd = sinf(ang);
    return {(s0 * a.w + s1 * b.w) / sd, (s0 * a.x + s1 * b.x) / sd, (s0 * a.y + s1 * b.y) / sd, (s0 * a.z + s1 * b.z) / sd};
}

struct ShutterPoses {  // start / end world->camera pose of one camera
    f3 t0, t1;
    quat q0, q1;
    GSX_DEV ShutterPoses(const float* __restrict__ vm0, const float* __restrict__ vm1) {
        q0 = quat_from_viewmat(vm0);
        t0 = {vm0[3], vm0[7], vm0[11]};
        if (vm1 == nullptr) { q1 = q0; t1 = t0; }
        else { q1 = quat_from_viewmat(vm1); t1 = {vm1[3], vm1[7], vm1[11]}; }
    }
    GSX_DEV void at(float rt, f3& t, quat& q) const {
        t = (1.f - rt) * t0 + rt * t1;
        q = quat_slerp(q0, q1, rt);
    }
};

// ---------------------------------------------------------------------------------------------
// Camera models.  KIND: 0 = perfect pinhole, 1 = OpenCV pinhole (radial/tangential/thin-prism),
// 2 = OpenCV fisheye.
// ---------------------------------------------------------------------------------------------
enum { CAM_PERFECT_PINHOLE = 0, CAM_OPENCV_PINHOLE = 1, CAM_OPENCV_FISHEYE = 2 };

GSX_DEV float poly_horner(const float* c, int n, float x) {
    float y = 0.f;
    for (int i = n - 1; i >= 0; --i) y = x * y + c[i];
    return y;
}

template <int KIND> struct Camera {
    float fx, fy, cx, cy;
    uint32_t W, H;
    int shutter;
    float k[6];      // radial (pinhole k1..k6 / fisheye k1..k4)
    float p[2];      // tangential
    float s[4];      // thin prism
    float fwd[5], dfwd[5], bwd[2], max_angle;  // fisheye only

    GSX_DEV Camera(const gsx_cameras& cams, uint32_t cid, uint32_t W_, uint32_t H_) {
        const float* K = cams.Ks + cid * 9;
        fx = K[0]; fy = K[4]; cx = K[2]; cy = K[5];
        W = W_; H = H_;
        shutter = cams.shutter;
        for (int i = 0; i < 6; ++i) k[i] = 0.f;
        p[0] = p[1] = 0.f;
        s[0] = s[1] = s[2] = s[3] = 0.f;
        if (KIND == CAM_OPENCV_PINHOLE) {
            if (cams.radial) for (int i = 0; i < 6; ++i) k[i] = cams.radial[cid * 6 + i];
            if (cams.tangential) for (int i = 0; i < 2; ++i) p[i] = cams.tangential[cid * 2 + i];
            if (cams.thin_prism) for (int i = 0; i < 4; ++i) s[i] = cams.thin_prism[cid * 4 + i];
        }
        if (KIND == CAM_OPENCV_FISHEYE) {
            if (cams.radial) for (int i = 0; i < 4; ++i) k[i] = cams.radial[cid * 4 + i];
            init_fisheye();
        }
    }

    GSX_DEV bool in_bounds(f2 ip, float mf) const {
        const float MX = (float)W * mf, MY = (float)H * mf;
        bool v = true;
        v &= (-MX) <= ip.x && ip.x < ((float)W + MX);
        v &= (-MY) <= ip.y && ip.y < ((float)H + MY);
        return v;
    }

    // ---- fisheye set-up: gsplat/Cameras.cuh:760-815, 833-884 ----
    GSX_DEV static float fisheye_max_angle(float a, float b, float c) {
        const float INF = 3.402823466e+38f;
        if (c == 0.f) {
            if (b == 0.f) return a >= 0.f ? INF : -1.f / a;
            float delta = a * a - 4.f * b;
            if (delta >= 0.f) { delta = sqrtf(delta) - a; if (delta > 0.f) return 2.f / delta; }
        } else {
            const float boc = b / c, boc2 = boc * boc;
            const float t1 = (9.f * a * boc - 2.f * b * boc2 - 27.f) / c;
            const float t2 = 3.f * a / c - boc2;
            const float delta = t1 * t1 + 4.f * t2 * t2 * t2;
            if (delta >= 0.f) {
                const float d2 = sqrtf(delta);
                const float cr = cbrtf((d2 + t1) / 2.f);
                if (cr != 0.f) { const float sol = (cr - (t2 / cr) - boc) / 3.f; if (sol > 0.f) return sol; }
            } else {
                const float theta = atan2f(sqrtf(-delta), t1) / 3.f;
                const float ttp = 2.f * 3.14159265358979323846f / 3.f;
                const float t3 = 2.f * sqrtf(-t2);
                float sol = INF;
                for (int i = -1; i <= 1; ++i) {
                    const float v = (t3 * cosf(theta + (float)i * ttp) - boc) / 3.f;
                    if (v > 0.f) sol = fminf(sol, v);
                }
                return sol;
            }
        }
        return INF;
    }
    GSX_DEV void init_fisheye() {
        fwd[0] = 1.f; fwd[1] = k[0]; fwd[2] = k[1]; fwd[3] = k[2]; fwd[4] = k[3];
        dfwd[0] = 1.f; dfwd[1] = 3.f * k[0]; dfwd[2] = 5.f * k[1]; dfwd[3] = 7.f * k[2]; dfwd[4] = 9.f * k[3];
        const float mdx = fmaxf((float)W - cx, cx), mdy = fmaxf((float)H - cy, cy);
        const float max_r = sqrtf(mdx * mdx + mdy * mdy);
        if (k[3] == 0.f) {
            max_angle = sqrtf(fisheye_max_angle(3.f * k[0], 5.f * k[1], 7.f * k[2]));
        } else {
            const float dd[4] = {6.f * k[0], 20.f * k[1], 42.f * k[2], 72.f * k[3]};
            bool conv = false;
            float x = 1.57f;
            for (int j = 0; j < 20; ++j) {
                const float dfdx = x * poly_horner(dd, 4, x * x);
                const float res = poly_horner(dfwd, 5, x * x);
                const float dx = res / dfdx;
                x -= dx;
                if (fabsf(dx) < 1e-6f) { conv = true; break; }
            }
            max_angle = x;
            if (!conv || max_angle <= 0.f) max_angle = 3.402823466e+38f;
        }
        max_angle = fminf(max_angle, fmaxf(max_r / fx, max_r / fy));
        const float mnd = fmaxf((float)W / 2.f / fx, (float)H / 2.f / fy);
        bwd[0] = 0.f;
        bwd[1] = max_angle / mnd;
    }

    GSX_DEV void distortion(f2 uv, float& icD, f2& delta) const {  // Cameras.cuh:504-533
        const float ux2 = uv.x * uv.x, uy2 = uv.y * uv.y;
        const float r2 = ux2 + uy2;
        const float a1 = 2.f * uv.x * uv.y, a2 = r2 + 2.f * ux2, a3 = r2 + 2.f * uy2;
        const float num = 1.f + r2 * (k[0] + r2 * (k[1] + r2 * k[2]));
        const float den = 1.f + r2 * (k[3] + r2 * (k[4] + r2 * k[5]));
        icD = num / den;
        delta.x = p[0] * a1 + p[1] * a2 + r2 * (s[0] + r2 * s[1]);
        delta.y = p[0] * a3 + p[1] * a1 + r2 * (s[2] + r2 * s[3]);
    }

    // camera-space point -> pixel (+validity incl. the image margin)
    GSX_DEV bool project(f3 r, float mf, f2& ip) const {
        ip = {0.f, 0.f};
        if (r.z <= 0.f) return false;
        if (KIND == CAM_PERFECT_PINHOLE) {
            ip = {(r.x / r.z) * fx + cx, (r.y / r.z) * fy + cy};
            return in_bounds(ip, mf);
        } else if (KIND == CAM_OPENCV_PINHOLE) {
            const f2 uvn{r.x / r.z, r.y / r.z};
            float icD; f2 d;
            distortion(uvn, icD, d);
            const bool valid_radial = icD > 0.8f;
            ip = {(icD * uvn.x + d.x) * fx + cx, (icD * uvn.y + d.y) * fy + cy};
            return valid_radial & in_bounds(ip, mf);
        } else {
            const float ax = fabsf(r.x), ay = fabsf(r.y);
            const float mn = fminf(ax, ay), mx = fmaxf(ax, ay);
            float nrm = 0.f;
            if (mx > 0.f) { const float q = mn / mx; nrm = mx * sqrtf(1.f + q * q); }
            if (nrm <= 0.f) nrm = 1.1920929e-7f;
            const float theta_full = atan2f(nrm, r.z);
            const float theta = theta_full < max_angle ? theta_full : max_angle;
            const float delta = theta * poly_horner(fwd, 5, theta * theta) / nrm;
            if (delta <= 0.f) return false;
            ip = {fx * delta * r.x + cx, fy * delta * r.y + cy};
            return in_bounds(ip, mf) & (theta <= max_angle);
        }
    }

    // pixel -> unit camera ray (+validity)
    GSX_DEV bool unproject(f2 ip, f3& dir) const {
        if (KIND == CAM_PERFECT_PINHOLE) {
            const f3 c{(ip.x - cx) / fx, (ip.y - cy) / fy, 1.f};
            const float len = sqrtf(dot3(c, c));
            dir = {c.x / len, c.y / len, c.z / len};
            return true;
        } else if (KIND == CAM_OPENCV_PINHOLE) {  // Newton undistortion, <=5 iterations, Cameras.cuh:698-754
            const float xd = (ip.x - cx) / fx, yd = (ip.y - cy) / fy;
            float x = xd, y = yd;
            bool converged = false;
            for (int it = 0; it < 5; ++it) {
                const float r = x * x + y * y, r2 = r * r;
                const float al = 1.f + r * (k[0] + r * (k[1] + r * k[2]));
                const float be = 1.f + r * (k[3] + r * (k[4] + r * k[5]));
                const float d = al / be;
                if (d <= 0.f) break;
                const float fxv = d * x + 2.f * p[0] * x * y + p[1] * (r + 2.f * x * x) + s[0] * r + s[1] * r2 - xd;
                const float fyv = d * y + 2.f * p[1] * x * y + p[0] * (r + 2.f * y * y) + s[2] * r + s[3] * r2 - yd;
                const float al_r = k[0] + r * (2.f * k[1] + r * (3.f * k[2]));
                const float be_r = k[3] + r * (2.f * k[4] + r * (3.f * k[5]));
                const float d_r = (al_r * be - al * be_r) / (be * be);
                const float d_x = 2.f * x * d_r, d_y = 2.f * y * d_r;
                const float fx_x = d + d_x * x + 2.f * p[0] * y + 6.f * p[1] * x + 2.f * x * (s[0] + 2.f * s[1] * r);
                const float fx_y = d_y * x + 2.f * p[0] * x + 2.f * p[1] * y + 2.f * y * (s[0] + 2.f * s[1] * r);
                const float fy_x = d_x * y + 2.f * p[1] * y + 2.f * p[0] * x + 2.f * x * (s[2] + 2.f * s[3] * r);
                const float fy_y = d + d_y * y + 2.f * p[1] * x + 6.f * p[0] * y + 2.f * y * (s[2] + 2.f * s[3] * r);
                const float det = fx_y * fy_x - fx_x * fy_y;
                if (fabsf(det) < 1e-6f) break;
                const float dx = (fxv * fy_y - fyv * fx_y) / det;
                const float dy = (fyv * fx_x - fxv * fy_x) / det;
                x += dx; y += dy;
                if (fabsf(dx) < 1e-6f && fabsf(dy) < 1e-6f) { converged = true; break; }
            }
            const f3 c{x, y, 1.f};
            const float len = sqrtf(dot3(c, c));
            dir = {c.x / len, c.y / len, c.z / len};
            return converged;
        } else {  // fisheye, Cameras.cuh:961-1000
            const f2 uv{(ip.x - cx) / fx, (ip.y - cy) / fy};
            const float delta = sqrtf(uv.x * uv.x + uv.y * uv.y);
            bool conv = false;
            float th = poly_horner(bwd, 2, delta);
            for (int j = 0; j < 20; ++j) {
                const float dfdx = poly_horner(dfwd, 5, th * th);
                const float res = th * poly_horner(fwd, 5, th * th) - delta;
                const float dx = res / dfdx;
                th -= dx;
                if (fabsf(dx) < 1e-6f) { conv = true; break; }
            }
            if (th < 0.f || th >= max_angle || !conv) { dir = {0.f, 0.f, 1.f}; return false; }
            if (delta >= 1e-6f) {
                const float sf = sinf(th) / delta;
                dir = {sf * uv.x, sf * uv.y, cosf(th)};
            } else {
                dir = {0.f, 0.f, 1.f};
            }
            return true;
        }
    }

    GSX_DEV float relative_frame_time(f2 ip) const {  // Cameras.cuh:293-320
        switch (shutter) {
        case GSX_SHUTTER_ROLLING_TOP_TO_BOTTOM: return floorf(ip.y) / (float)(H - 1);
        case GSX_SHUTTER_ROLLING_LEFT_TO_RIGHT: return floorf(ip.x) / (float)(W - 1);
        case GSX_SHUTTER_ROLLING_BOTTOM_TO_TOP: return ((float)H - ceilf(ip.y)) / (float)(H - 1);
        case GSX_SHUTTER_ROLLING_RIGHT_TO_LEFT: return ((float)W - ceilf(ip.x)) / (float)(W - 1);
        default: return 0.f;
        }
    }

    // world point -> pixel through the shutter pose(s): Cameras.cuh:346-413
    GSX_DEV bool world_to_image(f3 wp, const ShutterPoses& sp, float mf, f2& ip) const {
        f2 ps;
        const bool vs = project(quat_rotate(sp.q0, wp) + sp.t0, mf, ps);
        if (shutter == GSX_SHUTTER_GLOBAL) { ip = ps; return vs; }
        f2 pe;
        const bool ve = project(quat_rotate(sp.q1, wp) + sp.t1, mf, pe);
        f2 prev;
        if (vs) prev = ps;
        else if (ve) prev = pe;
        else { ip = pe; return false; }
        for (int j = 0; j < 10; ++j) {
            const float rt = relative_frame_time(prev);
            f3 t; quat q;
            sp.at(rt, t, q);
            f2 pr;
            project(quat_rotate(q, wp) + t, mf, pr);
            prev = pr;
        }
        ip = prev;
        return true;
    }

    // pixel -> world ray: Cameras.cuh:322-339, 261-265
    GSX_DEV bool pixel_to_world_ray(f2 ip, const ShutterPoses& sp, f3& org, f3& dir) const {
        f3 cdir;
        if (!unproject(ip, cdir)) { org = {0.f, 0.f, 0.f}; dir = {0.f, 0.f, 0.f}; return false; }
        f3 t; quat q;
        sp.at(relative_frame_time(ip), t, q);
        const m33 Rinv = quat_to_mat_raw(quat_conj_over_norm2(q));
        const f3 rt = mul(Rinv, t);
        org = {-rt.x, -rt.y, -rt.z};
        dir = mul(Rinv, cdir);
        return true;
    }
};

// ---------------------------------------------------------------------------------------------
// wave64 helpers
// ---------------------------------------------------------------------------------------------
// inclusive prefix sum over the 64 lanes with DPP row operations (no LDS traffic: six adds)
GSX_DEV uint32_t wave_incl_scan_u32(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true);   // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true);   // row_bcast:31 -> rows 2, 3
    return v;
}

// Sum over the 64 lanes of a wave using DPP row operations (no LDS traffic); result valid in lane 63.
GSX_DEV float wave_sum_to_lane63(float v) {
    // within rows of 16 lanes
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));  // row_shr:1
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));  // row_shr:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));  // row_shr:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));  // row_shr:8
    // across rows
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, true));  // row_bcast:15
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xc, 0xf, true));  // row_bcast:31
    return v;
}

}  // namespace gsx
