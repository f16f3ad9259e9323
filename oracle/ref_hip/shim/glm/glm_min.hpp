// TEST INFRASTRUCTURE ONLY (oracle/_ref build, see oracle/ref_hip/README.md).  Never included by the product.
//
// A minimal, builder-written subset of the glm interface (glm 1.0.x semantics; glm itself is an un-vendored vcpkg
// dependency of the reference, pinned only by vcpkg.json's builtin-baseline), just large enough for the reference's
// gsplat/*.cu{,h} to compile unmodified with hipcc.  Column-major matrices (m[c][r], mat(…) fills columns), quaternion
// constructor order (w, x, y, z), arithmetic written in the operand order glm documents / SURVEY.md Appendix B lists:
//   dot(vec3) = x+y+z left to right; dot(vec4 / quat) = (x+y)+(z+w) pairs; cross as (a.y b.z − b.y a.z, …);
//   mat*vec = m[0]*v.x + m[1]*v.y + …; mat*mat column by column; inverse(mat2) = adjugate × (1/det);
//   quat*vec3 = v + 2(w·uv + uuv); mat3_cast / quat_cast / slerp (lerp fallback at cosθ > 1−ε) / inverse(quat) = conj/dot.
#pragma once
#include <cmath>
#include <cstddef>
#include <limits>
#include <type_traits>

#if defined(__HIPCC__)
#define GLM_FUNC __host__ __device__ inline
#else
#define GLM_FUNC inline
#endif

namespace glm {

    typedef int length_t;
    enum qualifier { packed_highp,
                     defaultp = packed_highp };

    template <length_t L, typename T, qualifier Q = defaultp>
    struct vec;
    template <length_t C, length_t R, typename T, qualifier Q = defaultp>
    struct mat;
    template <typename T, qualifier Q = defaultp>
    struct qua;

    // ------------------------------------------------------------------ vectors
    template <typename T, qualifier Q>
    struct vec<2, T, Q> {
        T x, y;
        vec() = default;
        GLM_FUNC constexpr explicit vec(T s) : x(s),
                                      y(s) {}
        GLM_FUNC constexpr vec(T a, T b) : x(a),
                                           y(b) {}
        template <typename U>
        GLM_FUNC constexpr vec(vec<2, U, Q> const& v) : x(static_cast<T>(v.x)),
                                                        y(static_cast<T>(v.y)) {}
        GLM_FUNC T& operator[](length_t i) { return (&x)[i]; }
        GLM_FUNC T const& operator[](length_t i) const { return (&x)[i]; }
        GLM_FUNC vec& operator+=(vec const& v) {
            x += v.x;
            y += v.y;
            return *this;
        }
        GLM_FUNC vec& operator-=(vec const& v) {
            x -= v.x;
            y -= v.y;
            return *this;
        }
        GLM_FUNC vec& operator*=(T s) {
            x *= s;
            y *= s;
            return *this;
        }
        static GLM_FUNC constexpr length_t length() { return 2; }
    };
    template <typename T, qualifier Q>
    struct vec<3, T, Q> {
        T x, y, z;
        vec() = default;
        GLM_FUNC constexpr explicit vec(T s) : x(s),
                                      y(s),
                                      z(s) {}
        GLM_FUNC constexpr vec(T a, T b, T c) : x(a),
                                                y(b),
                                                z(c) {}
        template <typename U>
        GLM_FUNC constexpr vec(vec<3, U, Q> const& v) : x(static_cast<T>(v.x)),
                                                        y(static_cast<T>(v.y)),
                                                        z(static_cast<T>(v.z)) {}
        GLM_FUNC T& operator[](length_t i) { return (&x)[i]; }
        GLM_FUNC T const& operator[](length_t i) const { return (&x)[i]; }
        GLM_FUNC vec& operator+=(vec const& v) {
            x += v.x;
            y += v.y;
            z += v.z;
            return *this;
        }
        GLM_FUNC vec& operator-=(vec const& v) {
            x -= v.x;
            y -= v.y;
            z -= v.z;
            return *this;
        }
        GLM_FUNC vec& operator*=(T s) {
            x *= s;
            y *= s;
            z *= s;
            return *this;
        }
        static GLM_FUNC constexpr length_t length() { return 3; }
    };
    template <typename T, qualifier Q>
    struct vec<4, T, Q> {
        T x, y, z, w;
        vec() = default;
        GLM_FUNC constexpr explicit vec(T s) : x(s),
                                      y(s),
                                      z(s),
                                      w(s) {}
        GLM_FUNC constexpr vec(T a, T b, T c, T d) : x(a),
                                                     y(b),
                                                     z(c),
                                                     w(d) {}
        template <typename U>
        GLM_FUNC constexpr vec(vec<4, U, Q> const& v) : x(static_cast<T>(v.x)),
                                                        y(static_cast<T>(v.y)),
                                                        z(static_cast<T>(v.z)),
                                                        w(static_cast<T>(v.w)) {}
        GLM_FUNC T& operator[](length_t i) { return (&x)[i]; }
        GLM_FUNC T const& operator[](length_t i) const { return (&x)[i]; }
        GLM_FUNC vec& operator+=(vec const& v) {
            x += v.x;
            y += v.y;
            z += v.z;
            w += v.w;
            return *this;
        }
        GLM_FUNC vec& operator-=(vec const& v) {
            x -= v.x;
            y -= v.y;
            z -= v.z;
            w -= v.w;
            return *this;
        }
        GLM_FUNC vec& operator*=(T s) {
            x *= s;
            y *= s;
            z *= s;
            w *= s;
            return *this;
        }
        static GLM_FUNC constexpr length_t length() { return 4; }
    };

#define GLM_MIN_VEC_BINOP(OP)                                                                                       \
    template <typename T, qualifier Q>                                                                              \
    GLM_FUNC vec<2, T, Q> operator OP(vec<2, T, Q> const& a, vec<2, T, Q> const& b) { return {a.x OP b.x, a.y OP b.y}; } \
    template <typename T, qualifier Q>                                                                              \
    GLM_FUNC vec<3, T, Q> operator OP(vec<3, T, Q> const& a, vec<3, T, Q> const& b) {                               \
        return {a.x OP b.x, a.y OP b.y, a.z OP b.z};                                                                \
    }                                                                                                               \
    template <typename T, qualifier Q>                                                                              \
    GLM_FUNC vec<4, T, Q> operator OP(vec<4, T, Q> const& a, vec<4, T, Q> const& b) {                               \
        return {a.x OP b.x, a.y OP b.y, a.z OP b.z, a.w OP b.w};                                                    \
    }                                                                                                               \
    template <typename T, qualifier Q>                                                                              \
    GLM_FUNC vec<2, T, Q> operator OP(vec<2, T, Q> const& a, T s) { return {a.x OP s, a.y OP s}; }                  \
    template <typename T, qualifier Q>                                                                              \
    GLM_FUNC vec<3, T, Q> operator OP(vec<3, T, Q> const& a, T s) { return {a.x OP s, a.y OP s, a.z OP s}; }        \
    template <typename T, qualifier Q>                                                                              \
    GLM_FUNC vec<4, T, Q> operator OP(vec<4, T, Q> const& a, T s) { return {a.x OP s, a.y OP s, a.z OP s, a.w OP s}; } \
    template <typename T, qualifier Q>                                                                              \
    GLM_FUNC vec<2, T, Q> operator OP(T s, vec<2, T, Q> const& a) { return {s OP a.x, s OP a.y}; }                  \
    template <typename T, qualifier Q>                                                                              \
    GLM_FUNC vec<3, T, Q> operator OP(T s, vec<3, T, Q> const& a) { return {s OP a.x, s OP a.y, s OP a.z}; }        \
    template <typename T, qualifier Q>                                                                              \
    GLM_FUNC vec<4, T, Q> operator OP(T s, vec<4, T, Q> const& a) { return {s OP a.x, s OP a.y, s OP a.z, s OP a.w}; }
    GLM_MIN_VEC_BINOP(+)
    GLM_MIN_VEC_BINOP(-)
    GLM_MIN_VEC_BINOP(*)
    GLM_MIN_VEC_BINOP(/)
#undef GLM_MIN_VEC_BINOP

    template <typename T, qualifier Q>
    GLM_FUNC vec<2, T, Q> operator-(vec<2, T, Q> const& a) { return {-a.x, -a.y}; }
    template <typename T, qualifier Q>
    GLM_FUNC vec<3, T, Q> operator-(vec<3, T, Q> const& a) { return {-a.x, -a.y, -a.z}; }
    template <typename T, qualifier Q>
    GLM_FUNC vec<4, T, Q> operator-(vec<4, T, Q> const& a) { return {-a.x, -a.y, -a.z, -a.w}; }

    template <typename T, qualifier Q>
    GLM_FUNC T dot(vec<2, T, Q> const& a, vec<2, T, Q> const& b) {
        T tx = a.x * b.x, ty = a.y * b.y;
        return tx + ty;
    }
    template <typename T, qualifier Q>
    GLM_FUNC T dot(vec<3, T, Q> const& a, vec<3, T, Q> const& b) {
        T tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
        return tx + ty + tz;
    }
    template <typename T, qualifier Q>
    GLM_FUNC T dot(vec<4, T, Q> const& a, vec<4, T, Q> const& b) {
        T tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z, tw = a.w * b.w;
        return (tx + ty) + (tz + tw);
    }
    template <typename T, qualifier Q>
    GLM_FUNC vec<3, T, Q> cross(vec<3, T, Q> const& x, vec<3, T, Q> const& y) {
        return {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y};
    }
    template <length_t L, typename T, qualifier Q>
    GLM_FUNC T length(vec<L, T, Q> const& v) { return ::sqrt(dot(v, v)); }
    template <length_t L, typename T, qualifier Q>
    GLM_FUNC vec<L, T, Q> normalize(vec<L, T, Q> const& v) { return v * (T(1) / ::sqrt(dot(v, v))); }

    template <typename T>
    GLM_FUNC vec<2, T, defaultp> make_vec2(T const* p) { return {p[0], p[1]}; }
    template <typename T>
    GLM_FUNC vec<3, T, defaultp> make_vec3(T const* p) { return {p[0], p[1], p[2]}; }
    template <typename T>
    GLM_FUNC vec<4, T, defaultp> make_vec4(T const* p) { return {p[0], p[1], p[2], p[3]}; }

    // ------------------------------------------------------------------ matrices (C columns of R rows)
    template <length_t C, length_t R, typename T, qualifier Q>
    struct mat {
        typedef vec<R, T, Q> col_type;
        typedef vec<C, T, Q> row_type;
        col_type value[C];
        mat() = default;
        GLM_FUNC mat(T s) {  // diagonal
            for (length_t c = 0; c < C; ++c)
                for (length_t r = 0; r < R; ++r)
                    value[c][r] = (c == r) ? s : T(0);
        }
        template <length_t CC = C, length_t RR = R, typename = typename std::enable_if<CC == 2 && RR == 2>::type>
        GLM_FUNC constexpr mat(T x0, T y0, T x1, T y1) : value{col_type(x0, y0), col_type(x1, y1)} {}
        template <length_t CC = C, length_t RR = R, typename = typename std::enable_if<CC == 3 && RR == 3>::type>
        GLM_FUNC constexpr mat(T x0, T y0, T z0, T x1, T y1, T z1, T x2, T y2, T z2)
            : value{col_type(x0, y0, z0), col_type(x1, y1, z1), col_type(x2, y2, z2)} {}
        template <length_t CC = C, typename = typename std::enable_if<CC == 2>::type>
        GLM_FUNC constexpr mat(col_type const& c0, col_type const& c1) : value{c0, c1} {}
        template <length_t CC = C, typename = typename std::enable_if<CC == 3>::type>
        GLM_FUNC constexpr mat(col_type const& c0, col_type const& c1, col_type const& c2) : value{c0, c1, c2} {}
        template <length_t CC = C, typename = typename std::enable_if<CC == 4>::type>
        GLM_FUNC constexpr mat(col_type const& c0, col_type const& c1, col_type const& c2, col_type const& c3)
            : value{c0, c1, c2, c3} {}
        GLM_FUNC col_type& operator[](length_t i) { return value[i]; }
        GLM_FUNC col_type const& operator[](length_t i) const { return value[i]; }
        GLM_FUNC mat& operator+=(mat const& m) {
            for (length_t c = 0; c < C; ++c)
                value[c] += m.value[c];
            return *this;
        }
        GLM_FUNC mat& operator-=(mat const& m) {
            for (length_t c = 0; c < C; ++c)
                value[c] -= m.value[c];
            return *this;
        }
    };

    template <length_t C, length_t R, typename T, qualifier Q>
    GLM_FUNC mat<C, R, T, Q> operator+(mat<C, R, T, Q> const& a, mat<C, R, T, Q> const& b) {
        mat<C, R, T, Q> m;
        for (length_t c = 0; c < C; ++c)
            m[c] = a[c] + b[c];
        return m;
    }
    template <length_t C, length_t R, typename T, qualifier Q>
    GLM_FUNC mat<C, R, T, Q> operator-(mat<C, R, T, Q> const& a, mat<C, R, T, Q> const& b) {
        mat<C, R, T, Q> m;
        for (length_t c = 0; c < C; ++c)
            m[c] = a[c] - b[c];
        return m;
    }
    template <length_t C, length_t R, typename T, qualifier Q>
    GLM_FUNC mat<C, R, T, Q> operator-(mat<C, R, T, Q> const& a) {
        mat<C, R, T, Q> m;
        for (length_t c = 0; c < C; ++c)
            m[c] = -a[c];
        return m;
    }
    template <length_t C, length_t R, typename T, qualifier Q>
    GLM_FUNC mat<C, R, T, Q> operator*(mat<C, R, T, Q> const& a, T s) {
        mat<C, R, T, Q> m;
        for (length_t c = 0; c < C; ++c)
            m[c] = a[c] * s;
        return m;
    }
    template <length_t C, length_t R, typename T, qualifier Q>
    GLM_FUNC mat<C, R, T, Q> operator*(T s, mat<C, R, T, Q> const& a) {
        mat<C, R, T, Q> m;
        for (length_t c = 0; c < C; ++c)
            m[c] = a[c] * s;
        return m;
    }
    // matrix × column vector: m[0]*v[0] + m[1]*v[1] + … (left to right)
    template <length_t C, length_t R, typename T, qualifier Q>
    GLM_FUNC vec<R, T, Q> operator*(mat<C, R, T, Q> const& m, typename mat<C, R, T, Q>::row_type const& v) {  // (non-deduced, as in glm)
        vec<R, T, Q> r = m[0] * v[0];
        for (length_t c = 1; c < C; ++c)
            r = r + m[c] * v[c];
        return r;
    }
    // matrix × matrix, column by column
    template <length_t K, length_t R, length_t C2, typename T, qualifier Q>
    GLM_FUNC mat<C2, R, T, Q> operator*(mat<K, R, T, Q> const& a, mat<C2, K, T, Q> const& b) {
        mat<C2, R, T, Q> m;
        for (length_t c = 0; c < C2; ++c)
            m[c] = a * b[c];
        return m;
    }
    template <length_t C, length_t R, typename T, qualifier Q>
    GLM_FUNC mat<R, C, T, Q> transpose(mat<C, R, T, Q> const& a) {
        mat<R, C, T, Q> m;
        for (length_t c = 0; c < C; ++c)
            for (length_t r = 0; r < R; ++r)
                m[r][c] = a[c][r];
        return m;
    }
    // outerProduct(c, r) = c · rᵀ : column i = c * r[i]
    template <length_t DA, length_t DB, typename T, qualifier Q>
    GLM_FUNC mat<DB, DA, T, Q> outerProduct(vec<DA, T, Q> const& c, vec<DB, T, Q> const& r) {
        mat<DB, DA, T, Q> m;
        for (length_t i = 0; i < DB; ++i)
            m[i] = c * r[i];
        return m;
    }
    template <typename T, qualifier Q>
    GLM_FUNC mat<2, 2, T, Q> inverse(mat<2, 2, T, Q> const& m) {
        T ood = T(1) / (m[0][0] * m[1][1] - m[1][0] * m[0][1]);
        return mat<2, 2, T, Q>(+m[1][1] * ood, -m[0][1] * ood, -m[1][0] * ood, +m[0][0] * ood);
    }
    template <typename T, qualifier Q>
    GLM_FUNC mat<3, 3, T, Q> inverse(mat<3, 3, T, Q> const& m) {
        T ood = T(1) / (+m[0][0] * (m[1][1] * m[2][2] - m[2][1] * m[1][2]) - m[1][0] * (m[0][1] * m[2][2] - m[2][1] * m[0][2]) +
                        m[2][0] * (m[0][1] * m[1][2] - m[1][1] * m[0][2]));
        mat<3, 3, T, Q> inv;
        inv[0][0] = +(m[1][1] * m[2][2] - m[2][1] * m[1][2]) * ood;
        inv[1][0] = -(m[1][0] * m[2][2] - m[2][0] * m[1][2]) * ood;
        inv[2][0] = +(m[1][0] * m[2][1] - m[2][0] * m[1][1]) * ood;
        inv[0][1] = -(m[0][1] * m[2][2] - m[2][1] * m[0][2]) * ood;
        inv[1][1] = +(m[0][0] * m[2][2] - m[2][0] * m[0][2]) * ood;
        inv[2][1] = -(m[0][0] * m[2][1] - m[2][0] * m[0][1]) * ood;
        inv[0][2] = +(m[0][1] * m[1][2] - m[1][1] * m[0][2]) * ood;
        inv[1][2] = -(m[0][0] * m[1][2] - m[1][0] * m[0][2]) * ood;
        inv[2][2] = +(m[0][0] * m[1][1] - m[1][0] * m[0][1]) * ood;
        return inv;
    }

    // ------------------------------------------------------------------ quaternions: constructor (w, x, y, z), storage x y z w
    template <typename T, qualifier Q>
    struct qua {
        T x, y, z, w;
        qua() = default;
        GLM_FUNC constexpr qua(T w_, T x_, T y_, T z_) : x(x_),
                                                         y(y_),
                                                         z(z_),
                                                         w(w_) {}
    };
    template <typename T, qualifier Q>
    GLM_FUNC qua<T, Q> operator-(qua<T, Q> const& q) { return qua<T, Q>(-q.w, -q.x, -q.y, -q.z); }
    template <typename T, qualifier Q>
    GLM_FUNC qua<T, Q> operator+(qua<T, Q> const& a, qua<T, Q> const& b) { return qua<T, Q>(a.w + b.w, a.x + b.x, a.y + b.y, a.z + b.z); }
    template <typename T, qualifier Q>
    GLM_FUNC qua<T, Q> operator*(qua<T, Q> const& q, T s) { return qua<T, Q>(q.w * s, q.x * s, q.y * s, q.z * s); }
    template <typename T, qualifier Q>
    GLM_FUNC qua<T, Q> operator*(T s, qua<T, Q> const& q) { return q * s; }
    template <typename T, qualifier Q>
    GLM_FUNC qua<T, Q> operator/(qua<T, Q> const& q, T s) { return qua<T, Q>(q.w / s, q.x / s, q.y / s, q.z / s); }
    template <typename T, qualifier Q>
    GLM_FUNC T dot(qua<T, Q> const& a, qua<T, Q> const& b) {
        T tx = a.w * b.w, ty = a.x * b.x, tz = a.y * b.y, tw = a.z * b.z;
        return (tx + ty) + (tz + tw);
    }
    template <typename T, qualifier Q>
    GLM_FUNC T length(qua<T, Q> const& q) { return ::sqrt(dot(q, q)); }
    template <typename T, qualifier Q>
    GLM_FUNC qua<T, Q> normalize(qua<T, Q> const& q) {
        T len = length(q);
        if (len <= T(0))
            return qua<T, Q>(T(1), T(0), T(0), T(0));
        T ool = T(1) / len;
        return qua<T, Q>(q.w * ool, q.x * ool, q.y * ool, q.z * ool);
    }
    template <typename T, qualifier Q>
    GLM_FUNC qua<T, Q> conjugate(qua<T, Q> const& q) { return qua<T, Q>(q.w, -q.x, -q.y, -q.z); }
    template <typename T, qualifier Q>
    GLM_FUNC qua<T, Q> inverse(qua<T, Q> const& q) { return conjugate(q) / dot(q, q); }
    // q * v
    template <typename T, qualifier Q>
    GLM_FUNC vec<3, T, Q> operator*(qua<T, Q> const& q, vec<3, T, Q> const& v) {
        vec<3, T, Q> const u(q.x, q.y, q.z);
        vec<3, T, Q> const uv(cross(u, v));
        vec<3, T, Q> const uuv(cross(u, uv));
        return v + ((uv * q.w) + uuv) * T(2);
    }
    template <typename T, qualifier Q>
    GLM_FUNC vec<3, T, Q> rotate(qua<T, Q> const& q, vec<3, T, Q> const& v) { return q * v; }
    template <typename T, qualifier Q>
    GLM_FUNC mat<3, 3, T, Q> mat3_cast(qua<T, Q> const& q) {
        mat<3, 3, T, Q> m(T(1));
        T qxx(q.x * q.x), qyy(q.y * q.y), qzz(q.z * q.z), qxz(q.x * q.z), qxy(q.x * q.y), qyz(q.y * q.z), qwx(q.w * q.x), qwy(q.w * q.y),
            qwz(q.w * q.z);
        m[0][0] = T(1) - T(2) * (qyy + qzz);
        m[0][1] = T(2) * (qxy + qwz);
        m[0][2] = T(2) * (qxz - qwy);
        m[1][0] = T(2) * (qxy - qwz);
        m[1][1] = T(1) - T(2) * (qxx + qzz);
        m[1][2] = T(2) * (qyz + qwx);
        m[2][0] = T(2) * (qxz + qwy);
        m[2][1] = T(2) * (qyz - qwx);
        m[2][2] = T(1) - T(2) * (qxx + qyy);
        return m;
    }
    template <typename T, qualifier Q>
    GLM_FUNC qua<T, Q> quat_cast(mat<3, 3, T, Q> const& m) {
        T fx = m[0][0] - m[1][1] - m[2][2];
        T fy = m[1][1] - m[0][0] - m[2][2];
        T fz = m[2][2] - m[0][0] - m[1][1];
        T fw = m[0][0] + m[1][1] + m[2][2];
        int big = 0;
        T fb = fw;
        if (fx > fb) {
            fb = fx;
            big = 1;
        }
        if (fy > fb) {
            fb = fy;
            big = 2;
        }
        if (fz > fb) {
            fb = fz;
            big = 3;
        }
        T bv = ::sqrt(fb + T(1)) * T(0.5);
        T mult = T(0.25) / bv;
        switch (big) {
        case 0: return qua<T, Q>(bv, (m[1][2] - m[2][1]) * mult, (m[2][0] - m[0][2]) * mult, (m[0][1] - m[1][0]) * mult);
        case 1: return qua<T, Q>((m[1][2] - m[2][1]) * mult, bv, (m[0][1] + m[1][0]) * mult, (m[2][0] + m[0][2]) * mult);
        case 2: return qua<T, Q>((m[2][0] - m[0][2]) * mult, (m[0][1] + m[1][0]) * mult, bv, (m[1][2] + m[2][1]) * mult);
        default: return qua<T, Q>((m[0][1] - m[1][0]) * mult, (m[2][0] + m[0][2]) * mult, (m[1][2] + m[2][1]) * mult, bv);
        }
    }
    template <typename T>
    GLM_FUNC T mix(T a, T b, T t) { return a * (T(1) - t) + b * t; }
    template <typename T, qualifier Q>
    GLM_FUNC qua<T, Q> slerp(qua<T, Q> const& x, qua<T, Q> const& y, T a) {
        qua<T, Q> z = y;
        T c = dot(x, y);
        if (c < T(0)) {
            z = -y;
            c = -c;
        }
        if (c > T(1) - std::numeric_limits<T>::epsilon())
            return qua<T, Q>(mix(x.w, z.w, a), mix(x.x, z.x, a), mix(x.y, z.y, a), mix(x.z, z.z, a));
        T angle = ::acos(c);
        return (::sin((T(1) - a) * angle) * x + ::sin(a * angle) * z) / ::sin(angle);
    }

    // ------------------------------------------------------------------ the aliases the reference uses
    typedef vec<2, float> vec2;
    typedef vec<3, float> vec3;
    typedef vec<4, float> vec4;
    typedef vec<2, float> fvec2;
    typedef vec<3, float> fvec3;
    typedef vec<4, float> fvec4;
    typedef mat<2, 2, float> mat2;
    typedef mat<3, 3, float> mat3;
    typedef mat<4, 4, float> mat4;
    typedef mat<2, 2, float> fmat2;
    typedef mat<3, 3, float> fmat3;
    typedef mat<4, 4, float> fmat4;
    typedef qua<float> fquat;
    typedef qua<float> quat;

}  // namespace glm
