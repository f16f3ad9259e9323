// gsx_sh.hip — spherical-harmonics colour evaluation, forward and backward, for gfx950.
//
// Replaces gsplat::spherical_harmonics_{fwd,bwd} (reference: gsplat/SphericalHarmonics.cpp:15-75,
// kernels gsplat/SphericalHarmonicsCUDA.cu:373-399 / 444-481, math :20-371; basis = Sloan,
// "Efficient Spherical Harmonic Evaluation", JCGT 2013).
//
// Design (HBM-bound streaming op, 217 B/Gaussian fwd and 409 B/Gaussian bwd at degree 3):
//   * one lane per Gaussian computes all three channels (the reference uses one thread per
//     (Gaussian, channel) and re-normalises the direction three times);
//   * the [K,3] coefficient rows of a wave's 64 Gaussians are one contiguous 64*K*12 B span of
//     HBM: the wave streams that span with fully coalesced 16 B/lane loads into LDS (odd row
//     stride -> conflict-free per-lane row reads) instead of 64 strided 192 B gathers;
//   * only the (deg+1)^2 active bases are read; masked Gaussians' rows are not read at all;
//   * backward writes v_coeffs rows (incl. the zeros above the active degree and for masked
//     Gaussians) through the same LDS tile with coalesced stores, so no separate memset pass
//     (the reference does at::zeros_like + a partial write).
#include <cstdlib>

#include "gsx_device.hpp"
#include "gsx_sh_basis.hpp"

namespace gsx {

constexpr int SH_BLOCK = 256;

// Stage the active part (nb3 floats) of the coefficient rows of this wave's 64 elements into LDS.
// `tile` points at the wave's LDS region, row stride `ls` floats (odd).
GSX_DEV void sh_stage_rows(const float* __restrict__ coeffs, uint32_t n, uint32_t e0, uint32_t K3, uint32_t nb3,
                           unsigned long long live, float* tile, uint32_t ls, uint32_t lane) {
    const uint32_t rows = min(64u, n - e0);
    if (nb3 == K3 && (K3 & 3u) == 0u) {
        // rows are back to back: stream float4s (16 B per lane, 1 KiB per wave instruction)
        const uint32_t q_per_row = K3 >> 2;
        const uint32_t total = rows * q_per_row;
        const float4* src = reinterpret_cast<const float4*>(coeffs + (size_t)e0 * K3);
        // batches of 6 loads per lane issued back to back (6 KiB in flight per wave) before the first LDS write
        constexpr int B = 6;
        for (uint32_t j0 = lane; j0 < total; j0 += 64 * B) {
            float4 v[B];
            uint32_t off[B];
            bool ok[B];
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const uint32_t j = j0 + 64u * b;
                const uint32_t e = j / q_per_row;
                off[b] = e * ls + ((j - e * q_per_row) << 2);
                ok[b] = j < total && ((live >> (e & 63u)) & 1ull);
                if (ok[b]) v[b] = src[j];
            }
#pragma unroll
            for (int b = 0; b < B; ++b)
                if (ok[b]) {
                    float* d = tile + off[b];
                    d[0] = v[b].x; d[1] = v[b].y; d[2] = v[b].z; d[3] = v[b].w;
                }
        }
    } else {
        const uint32_t total = rows * nb3;
        for (uint32_t j = lane; j < total; j += 64) {
            const uint32_t e = j / nb3, r = j - e * nb3;
            if (!((live >> e) & 1ull)) continue;
            tile[e * ls + r] = coeffs[(size_t)(e0 + e) * K3 + r];
        }
    }
}

template <int DEG>
__global__ __launch_bounds__(SH_BLOCK) void sh_fwd_kernel(uint32_t n, uint32_t K, const float* __restrict__ dirs,
                                                          const float* __restrict__ coeffs,
                                                          const uint8_t* __restrict__ masks,
                                                          float* __restrict__ colors) {
    constexpr int NB = (DEG + 1) * (DEG + 1);
    constexpr uint32_t NB3 = NB * 3;
    constexpr uint32_t LS = NB3 | 1u;
    extern __shared__ __attribute__((aligned(16))) float sh_lds[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t e0 = blockIdx.x * SH_BLOCK + wave * 64u;
    const uint32_t e = e0 + lane;
    const bool live = e < n && (masks == nullptr || masks[e] != 0);
    const unsigned long long live_mask = __builtin_amdgcn_ballot_w64(live);
    float* tile = sh_lds + wave * 64u * LS;
    if (e0 < n) sh_stage_rows(coeffs, n, e0, K * 3u, NB3, live_mask, tile, LS, lane);
    __syncthreads();
    if (!live) return;
    float x = dirs[(size_t)e * 3], y = dirs[(size_t)e * 3 + 1], z = dirs[(size_t)e * 3 + 2];
    if (DEG >= 1) {
        const float inorm = rsqrtf(x * x + y * y + z * z);
        x *= inorm; y *= inorm; z *= inorm;
    }
    float Y[NB];
    ShBasis<DEG>::template eval<false>(x, y, z, Y, nullptr, nullptr, nullptr);
    const float* row = tile + lane * LS;
    float r = 0.f, g = 0.f, b = 0.f;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        r += Y[k] * row[k * 3];
        g += Y[k] * row[k * 3 + 1];
        b += Y[k] * row[k * 3 + 2];
    }
    colors[(size_t)e * 3] = r;
    colors[(size_t)e * 3 + 1] = g;
    colors[(size_t)e * 3 + 2] = b;
}

template <int DEG, bool VDIRS>
__global__ __launch_bounds__(SH_BLOCK) void sh_bwd_kernel(uint32_t n, uint32_t K, const float* __restrict__ dirs,
                                                          const float* __restrict__ coeffs,
                                                          const uint8_t* __restrict__ masks,
                                                          const float* __restrict__ v_colors,
                                                          float* __restrict__ v_coeffs, float* __restrict__ v_dirs) {
    constexpr int NB = (DEG + 1) * (DEG + 1);
    constexpr uint32_t NB3 = NB * 3;
    extern __shared__ __attribute__((aligned(16))) float sh_lds[];
    const uint32_t K3 = K * 3u;
    const uint32_t LS = K3 | 1u;  // the tile holds full output rows
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t e0 = blockIdx.x * SH_BLOCK + wave * 64u;
    const uint32_t e = e0 + lane;
    const bool live = e < n && (masks == nullptr || masks[e] != 0);
    float* tile = sh_lds + wave * 64u * LS;
    float* row = tile + lane * LS;
    if (VDIRS && DEG >= 1 && e0 < n) {
        const unsigned long long live_mask = __builtin_amdgcn_ballot_w64(live);
        sh_stage_rows(coeffs, n, e0, K3, NB3, live_mask, tile, LS, lane);
    }
    __syncthreads();
    float vx = 0.f, vy = 0.f, vz = 0.f, inorm = 1.f;
    float x = 0.f, y = 0.f, z = 1.f;
    float Y[NB], Yx[NB], Yy[NB], Yz[NB];
    float vr = 0.f, vg = 0.f, vb = 0.f;
    if (live) {
        x = dirs[(size_t)e * 3]; y = dirs[(size_t)e * 3 + 1]; z = dirs[(size_t)e * 3 + 2];
        if (DEG >= 1) {
            inorm = rsqrtf(x * x + y * y + z * z);
            x *= inorm; y *= inorm; z *= inorm;
        }
        vr = v_colors[(size_t)e * 3]; vg = v_colors[(size_t)e * 3 + 1]; vb = v_colors[(size_t)e * 3 + 2];
        ShBasis<DEG>::template eval<VDIRS>(x, y, z, Y, Yx, Yy, Yz);
        if (VDIRS && DEG >= 1) {
#pragma unroll
            for (int k = 1; k < NB; ++k) {
                const float gk = row[k * 3] * vr + row[k * 3 + 1] * vg + row[k * 3 + 2] * vb;
                vx += Yx[k] * gk; vy += Yy[k] * gk; vz += Yz[k] * gk;
            }
        }
    }
    __syncthreads();  // all reads of the staged coefficients are done; reuse the tile for the output rows
    if (live) {
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            row[k * 3] = Y[k] * vr; row[k * 3 + 1] = Y[k] * vg; row[k * 3 + 2] = Y[k] * vb;
        }
        for (uint32_t j = NB3; j < K3; ++j) row[j] = 0.f;
    } else if (e < n) {
        for (uint32_t j = 0; j < K3; ++j) row[j] = 0.f;
    }
    __syncthreads();
    if (e0 < n) {   // coalesced store of the wave's rows*K3 contiguous output floats
        const uint32_t rows = min(64u, n - e0);
        float* dst = v_coeffs + (size_t)e0 * K3;
        if ((K3 & 3u) == 0u) {
            const uint32_t q_per_row = K3 >> 2, total = rows * q_per_row;
            for (uint32_t j = lane; j < total; j += 64) {
                const uint32_t er = j / q_per_row, r = (j - er * q_per_row) << 2;
                const float* s = tile + er * LS + r;
                reinterpret_cast<float4*>(dst)[j] = make_float4(s[0], s[1], s[2], s[3]);
            }
        } else {
            const uint32_t total = rows * K3;
            for (uint32_t j = lane; j < total; j += 64) {
                const uint32_t er = j / K3, r = j - er * K3;
                dst[j] = tile[er * LS + r];
            }
        }
    }
    if (VDIRS && e < n) {
        float ox = 0.f, oy = 0.f, oz = 0.f;
        if (live && DEG >= 1) {  // through the normalisation d/|d|
            const float d = vx * x + vy * y + vz * z;
            ox = (vx - d * x) * inorm; oy = (vy - d * y) * inorm; oz = (vz - d * z) * inorm;
        }
        v_dirs[(size_t)e * 3] = ox; v_dirs[(size_t)e * 3 + 1] = oy; v_dirs[(size_t)e * 3 + 2] = oz;
    }
}

void set_error(const char* msg);

// ------------------------------------------------------------------------------------------------
// Fused glue kernels (extensions, not part of gsplat/Ops.h): they replace the chains of small torch ops the
// reference's glue runs around the SH op every frame (src/training/rasterization/rasterizer.cpp:250-266):
//   campos = inverse(viewmat)[:3,3]; dirs = means - campos; masks = (radii > 0).all(-1);
//   colors = clamp_min(SH(dirs, coeffs, masks) + 0.5, 0)
// and their autograd backward (clamp mask, SH backward, dirs -> means, sum over cameras of the broadcast coeffs).
// One lane per Gaussian, all C cameras in a loop so the coefficient rows are staged through LDS once.
// campos is computed as -R^T t (exact for a rigid world->camera transform; upstream uses torch::inverse).
// ------------------------------------------------------------------------------------------------

template <int DEG>
__global__ __launch_bounds__(SH_BLOCK) void sh_colors_fwd_kernel(uint32_t C, uint32_t N, uint32_t K,
                                                                 const float* __restrict__ means,
                                                                 const float* __restrict__ viewmats,
                                                                 const float* __restrict__ coeffs,
                                                                 const int32_t* __restrict__ radii,
                                                                 float* __restrict__ colors) {
    constexpr int NB = (DEG + 1) * (DEG + 1);
    constexpr uint32_t NB3 = NB * 3;
    constexpr uint32_t LS = NB3 | 1u;
    extern __shared__ __attribute__((aligned(16))) float sh_lds[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t e0 = blockIdx.x * SH_BLOCK + wave * 64u;
    const uint32_t e = e0 + lane;
    bool any_live = false;
    if (e < N)
        for (uint32_t c = 0; c < C; ++c) {
            const int2 r = reinterpret_cast<const int2*>(radii)[(size_t)c * N + e];
            any_live |= (r.x > 0 && r.y > 0);
        }
    const unsigned long long live_mask = __builtin_amdgcn_ballot_w64(any_live);
    float* tile = sh_lds + wave * 64u * LS;
    if (e0 < N) sh_stage_rows(coeffs, N, e0, K * 3u, NB3, live_mask, tile, LS, lane);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the tile is private to this wave: no block barrier
    if (e >= N) return;
    if (!any_live) {  // masked for every camera: zero rows (the output buffer needs no pre-fill)
        for (uint32_t c = 0; c < C; ++c) {
            float* o = colors + ((size_t)c * N + e) * 3;
            o[0] = 0.f; o[1] = 0.f; o[2] = 0.f;
        }
        return;
    }
    const f3 mu{means[(size_t)e * 3], means[(size_t)e * 3 + 1], means[(size_t)e * 3 + 2]};
    const float* row = tile + lane * LS;
    for (uint32_t c = 0; c < C; ++c) {
        const int2 r = reinterpret_cast<const int2*>(radii)[(size_t)c * N + e];
        if (!(r.x > 0 && r.y > 0)) {  // masked: zero row
            float* o = colors + ((size_t)c * N + e) * 3;
            o[0] = 0.f; o[1] = 0.f; o[2] = 0.f;
            continue;
        }
        const f3 cp = cam_position(viewmats + c * 16);
        float x = mu.x - cp.x, y = mu.y - cp.y, z = mu.z - cp.z;
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
        float* o = colors + ((size_t)c * N + e) * 3;
        o[0] = fmaxf(cr + 0.5f, 0.f); o[1] = fmaxf(cg + 0.5f, 0.f); o[2] = fmaxf(cb + 0.5f, 0.f);
    }
}

// The same without the LDS tile: every lane reads the active part of ITS coefficient row straight into registers (NB3 / 4 loads of
// 16 B; the 64 rows of a wave are one contiguous span, every fetched line is consumed by the wave's own loads through L1).  No LDS ->
// the occupancy is set by the registers, not by 12.5 KB of tile per wave.  Needs (K * 3) % 4 == 0 (16 B aligned rows).
template <int DEG>
__global__ __launch_bounds__(SH_BLOCK) void sh_colors_fwd_direct_kernel(uint32_t C, uint32_t N, uint32_t K,
                                                                        const float* __restrict__ means,
                                                                        const float* __restrict__ viewmats,
                                                                        const float* __restrict__ coeffs,
                                                                        const int32_t* __restrict__ radii,
                                                                        float* __restrict__ colors) {
    constexpr int NB = (DEG + 1) * (DEG + 1);
    constexpr int NB3 = NB * 3;
    constexpr int NQ = (NB3 + 3) / 4;
    const uint32_t e = blockIdx.x * SH_BLOCK + threadIdx.x;
    if (e >= N) return;
    bool any_live = false;
    for (uint32_t c = 0; c < C; ++c) {
        const int2 r = reinterpret_cast<const int2*>(radii)[(size_t)c * N + e];
        any_live |= (r.x > 0 && r.y > 0);
    }
    if (!any_live) {
        for (uint32_t c = 0; c < C; ++c) {
            float* o = colors + ((size_t)c * N + e) * 3;
            o[0] = 0.f; o[1] = 0.f; o[2] = 0.f;
        }
        return;
    }
    float row[NQ * 4];
    const float4* src = reinterpret_cast<const float4*>(coeffs + (size_t)e * K * 3u);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const float4 v = src[q];
        row[q * 4] = v.x; row[q * 4 + 1] = v.y; row[q * 4 + 2] = v.z; row[q * 4 + 3] = v.w;
    }
    const f3 mu{means[(size_t)e * 3], means[(size_t)e * 3 + 1], means[(size_t)e * 3 + 2]};
    for (uint32_t c = 0; c < C; ++c) {
        const int2 r = reinterpret_cast<const int2*>(radii)[(size_t)c * N + e];
        float* o = colors + ((size_t)c * N + e) * 3;
        if (!(r.x > 0 && r.y > 0)) { o[0] = 0.f; o[1] = 0.f; o[2] = 0.f; continue; }
        const f3 cp = cam_position(viewmats + c * 16);
        float x = mu.x - cp.x, y = mu.y - cp.y, z = mu.z - cp.z;
        if (DEG >= 1) {
            const float inorm = rsqrtf(x * x + y * y + z * z);
            x *= inorm; y *= inorm; z *= inorm;
        }
        float Y[NB];
        ShBasis<DEG>::template eval<false>(x, y, z, Y, nullptr, nullptr, nullptr);
        float cr = 0.f, cg = 0.f, cb = 0.f;
#pragma unroll
        for (int k2 = 0; k2 < NB; ++k2) {
            cr += Y[k2] * row[k2 * 3];
            cg += Y[k2] * row[k2 * 3 + 1];
            cb += Y[k2] * row[k2 * 3 + 2];
        }
        o[0] = fmaxf(cr + 0.5f, 0.f); o[1] = fmaxf(cg + 0.5f, 0.f); o[2] = fmaxf(cb + 0.5f, 0.f);
    }
}

// Adam state and step of the SH tensor for the fused "SH backward + optimizer" launch (gsx_sh_colors_bwd_adam): the two column blocks
// sh0 (first 3 floats of a row) / shN keep their own step size and enable flag, as gsx_adam_step_split.
struct ShAdam {
    float* exp_avg; float* exp_avg_sq;
    float step0, stepN; int do0, doN;      // step = lr * bias_correction1_rcp
    float beta1, beta2, eps, bc2_sqrt_rcp;
};

// v_coeffs [N,K,3] is fully written (sum over cameras); v_means_inout [N,3] += d(colors)/d(means).
// ADAM: the gradient rows are not written; the wave's coalesced pass over its rows applies the Adam step to coeffs / exp_avg / exp_avg_sq
// in place instead (coeffs is then an in/out argument: `coeffs_rw`), which saves the 192 MB write + read of the SH gradient at 1 M Gaussians.
template <int DEG, bool ADAM>
__global__ __launch_bounds__(SH_BLOCK) void sh_colors_bwd_kernel(uint32_t C, uint32_t N, uint32_t K,
                                                                 const float* __restrict__ means,
                                                                 const float* __restrict__ viewmats,
                                                                 const float* coeffs,
                                                                 const int32_t* __restrict__ radii,
                                                                 const float* __restrict__ colors,
                                                                 const float* __restrict__ v_colors,
                                                                 float* v_coeffs,
                                                                 const float* __restrict__ v_means_in,
                                                                 float* __restrict__ v_means_out, ShAdam ad) {
    constexpr int NB = (DEG + 1) * (DEG + 1);
    constexpr uint32_t NB3 = NB * 3;
    extern __shared__ __attribute__((aligned(16))) float sh_lds[];
    const uint32_t K3 = K * 3u;
    const uint32_t LS = K3 | 1u;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t e0 = blockIdx.x * SH_BLOCK + wave * 64u;
    const uint32_t e = e0 + lane;
    // radii == nullptr: "pre-masked" colour gradients (multi-GPU colour exchange, distributed.ColorGradExchange): v_colors [C,N,3] already
    // carry the visibility and clamp masks of their camera (zero rows where a camera does not see the Gaussian), colors is not read
    const bool premasked = radii == nullptr;
    bool any_live = premasked && e < N;
    if (e < N && !premasked)
        for (uint32_t c = 0; c < C; ++c) {
            const int2 r = reinterpret_cast<const int2*>(radii)[(size_t)c * N + e];
            any_live |= (r.x > 0 && r.y > 0);
        }
    float* tile = sh_lds + wave * 64u * LS;
    float* row = tile + lane * LS;
    if (DEG >= 1 && e0 < N) {
        const unsigned long long live_mask = __builtin_amdgcn_ballot_w64(any_live);
        sh_stage_rows(coeffs, N, e0, K3, NB3, live_mask, tile, LS, lane);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the tile is private to this wave: no block barrier
    float vc[NB3];
#pragma unroll
    for (int k = 0; k < (int)NB3; ++k) vc[k] = 0.f;
    float gx = 0.f, gy = 0.f, gz = 0.f;
    if (any_live) {
        const f3 mu{means[(size_t)e * 3], means[(size_t)e * 3 + 1], means[(size_t)e * 3 + 2]};
        for (uint32_t c = 0; c < C; ++c) {
            const size_t ce = (size_t)c * N + e;
            float vr, vg, vb;
            if (premasked) {
                vr = v_colors[ce * 3]; vg = v_colors[ce * 3 + 1]; vb = v_colors[ce * 3 + 2];
                if (vr == 0.f && vg == 0.f && vb == 0.f) continue;   // this camera does not see (or clamps) the Gaussian
            } else {
                const int2 r = reinterpret_cast<const int2*>(radii)[ce];
                if (!(r.x > 0 && r.y > 0)) continue;
                // clamp_min(x + 0.5, 0) passes the gradient where the output is positive
                vr = colors[ce * 3] > 0.f ? v_colors[ce * 3] : 0.f;
                vg = colors[ce * 3 + 1] > 0.f ? v_colors[ce * 3 + 1] : 0.f;
                vb = colors[ce * 3 + 2] > 0.f ? v_colors[ce * 3 + 2] : 0.f;
            }
            const f3 cp = cam_position(viewmats + c * 16);
            float x = mu.x - cp.x, y = mu.y - cp.y, z = mu.z - cp.z, inorm = 1.f;
            if (DEG >= 1) {
                inorm = rsqrtf(x * x + y * y + z * z);
                x *= inorm; y *= inorm; z *= inorm;
            }
            float Y[NB], Yx[NB], Yy[NB], Yz[NB];
            ShBasis<DEG>::template eval<(DEG >= 1)>(x, y, z, Y, Yx, Yy, Yz);
            float vx = 0.f, vy = 0.f, vz = 0.f;
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                vc[k * 3] += Y[k] * vr; vc[k * 3 + 1] += Y[k] * vg; vc[k * 3 + 2] += Y[k] * vb;
                if (DEG >= 1 && k >= 1) {
                    const float gk = row[k * 3] * vr + row[k * 3 + 1] * vg + row[k * 3 + 2] * vb;
                    vx += Yx[k] * gk; vy += Yy[k] * gk; vz += Yz[k] * gk;
                }
            }
            if (DEG >= 1) {
                const float d = vx * x + vy * y + vz * z;
                gx += (vx - d * x) * inorm; gy += (vy - d * y) * inorm; gz += (vz - d * z) * inorm;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the tile is private to this wave: no block barrier  // staged coefficients fully consumed: reuse the tile for the output rows
    if (e < N) {
#pragma unroll
        for (int k = 0; k < (int)NB3; ++k) row[k] = vc[k];
        for (uint32_t j = NB3; j < K3; ++j) row[j] = 0.f;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the tile is private to this wave: no block barrier
    if (e0 < N) {
        const uint32_t rows = min(64u, N - e0);
        float* dst = v_coeffs + (size_t)e0 * K3;
        if ((K3 & 3u) == 0u) {
            const uint32_t q_per_row = K3 >> 2, total = rows * q_per_row;
            constexpr int B = 6;  // LDS reads of a batch overlap; the stores are fire-and-forget
            if (ADAM) {
                // the same coalesced pass, as an optimizer step: gradient from the tile, parameter / moments streamed (the parameter rows
                // were read a moment ago by this wave: the re-read is served by L2 / Infinity Cache)
                float4* pp = reinterpret_cast<float4*>(dst);   // dst == coeffs rows of this wave
                float4* pm = reinterpret_cast<float4*>(ad.exp_avg + (size_t)e0 * K3);
                float4* pv = reinterpret_cast<float4*>(ad.exp_avg_sq + (size_t)e0 * K3);
                constexpr int BA = 3;
                for (uint32_t j0 = lane; j0 < total; j0 += 64 * BA) {
                    float4 g[BA], p[BA], m[BA], v[BA];
                    uint32_t col[BA];
#pragma unroll
                    for (int b = 0; b < BA; ++b) {
                        const uint32_t j = min(j0 + 64u * b, total - 1u);
                        const uint32_t er = j / q_per_row, rr = (j - er * q_per_row) << 2;
                        const float* sp = tile + er * LS + rr;
                        g[b] = make_float4(sp[0], sp[1], sp[2], sp[3]);
                        col[b] = rr;
                        p[b] = pp[j]; m[b] = nt_load4(pm + j); v[b] = nt_load4(pv + j);   // moments: touched once per step, streamed
                    }
#pragma unroll
                    for (int b = 0; b < BA; ++b) {
                        if (j0 + 64u * b >= total) continue;
#define GSX_SH_ADAM1(F, J)                                                                                   \
                        {                                                                                    \
                            const bool a0 = col[b] + J < 3u;                                                 \
                            if (a0 ? ad.do0 : ad.doN) {                                                      \
                                m[b].F = ad.beta1 * m[b].F + (1.0f - ad.beta1) * g[b].F;                     \
                                v[b].F = ad.beta2 * v[b].F + (1.0f - ad.beta2) * g[b].F * g[b].F;            \
                                p[b].F -= (a0 ? ad.step0 : ad.stepN) * m[b].F / (sqrtf(v[b].F) * ad.bc2_sqrt_rcp + ad.eps); \
                            }                                                                                \
                        }
                        GSX_SH_ADAM1(x, 0) GSX_SH_ADAM1(y, 1) GSX_SH_ADAM1(z, 2) GSX_SH_ADAM1(w, 3)
#undef GSX_SH_ADAM1
                        const uint32_t j = j0 + 64u * b;
                        pp[j] = p[b]; nt_store4(m[b], pm + j); nt_store4(v[b], pv + j);   // (the parameters stay cached: the next render reads them)
                    }
                }
            } else
            for (uint32_t j0 = lane; j0 < total; j0 += 64 * B) {
                float4 v[B];
#pragma unroll
                for (int b = 0; b < B; ++b) {
                    const uint32_t j = min(j0 + 64u * b, total - 1u);
                    const uint32_t er = j / q_per_row, rr = (j - er * q_per_row) << 2;
                    const float* sp = tile + er * LS + rr;
                    v[b] = make_float4(sp[0], sp[1], sp[2], sp[3]);
                }
#pragma unroll
                for (int b = 0; b < B; ++b)
                    if (j0 + 64u * b < total) reinterpret_cast<float4*>(dst)[j0 + 64u * b] = v[b];
            }
        } else {
            const uint32_t total = rows * K3;
            for (uint32_t j = lane; j < total; j += 64) {
                const uint32_t er = j / K3, rr = j - er * K3;
                dst[j] = tile[er * LS + rr];
            }
        }
    }
    if (e < N) {
        const float bx = v_means_in ? v_means_in[(size_t)e * 3] : 0.f, by = v_means_in ? v_means_in[(size_t)e * 3 + 1] : 0.f,
                    bz = v_means_in ? v_means_in[(size_t)e * 3 + 2] : 0.f;
        v_means_out[(size_t)e * 3] = bx + gx; v_means_out[(size_t)e * 3 + 1] = by + gy; v_means_out[(size_t)e * 3 + 2] = bz + gz;
    }
}

// SplatData activations (reference: src/core/splat_data.cpp:267-286): scales = exp(raw) * modifier,
// quats = raw / max(|raw|, 1e-12) (torch normalize), opacities = sigmoid(raw); and their backward.
__global__ __launch_bounds__(256) void splat_activations_fwd_kernel(uint32_t N, const float* __restrict__ scaling_raw,
                                                                    const float* __restrict__ rotation_raw,
                                                                    const float* __restrict__ opacity_raw, float* __restrict__ scales,
                                                                    float* __restrict__ quats, float* __restrict__ opacities) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= N) return;
#pragma unroll
    for (int k = 0; k < 3; ++k) scales[(size_t)i * 3 + k] = expf(scaling_raw[(size_t)i * 3 + k]);
    const float4 q = reinterpret_cast<const float4*>(rotation_raw)[i];
    const float inv = 1.f / fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
    reinterpret_cast<float4*>(quats)[i] = make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
    opacities[i] = 1.f / (1.f + expf(-opacity_raw[i]));
}

__global__ __launch_bounds__(256) void splat_activations_bwd_kernel(uint32_t N, const float* __restrict__ scaling_raw,
                                                                    const float* __restrict__ rotation_raw,
                                                                    const float* __restrict__ opacity_raw,
                                                                    const float* __restrict__ v_scales, const float* __restrict__ v_quats,
                                                                    const float* __restrict__ v_opacities,
                                                                    float* __restrict__ v_scaling_raw, float* __restrict__ v_rotation_raw,
                                                                    float* __restrict__ v_opacity_raw, float scale_reg, float opacity_reg) {
    // scale_reg / opacity_reg != 0: the gradients of the MCMC strategy's regularisers scale_reg * mean(exp(s_raw)) and opacity_reg *
    // mean(sigmoid(o_raw)) (trainer.cpp:103-127 adds them to the loss) are added where exp(s) and sigmoid'(o) are formed anyway —
    // the caller passes reg / numel — instead of six elementwise launches behind the backward
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= N) return;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float e = expf(scaling_raw[(size_t)i * 3 + k]);
        float g = v_scales[(size_t)i * 3 + k] * e;
        if (scale_reg != 0.f) g = fmaf(scale_reg, e, g);
        v_scaling_raw[(size_t)i * 3 + k] = g;
    }
    const float4 q = reinterpret_cast<const float4*>(rotation_raw)[i];
    const float4 g = reinterpret_cast<const float4*>(v_quats)[i];
    const float nrm = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    float4 o;
    if (nrm > 1e-12f) {
        const float inv = 1.f / nrm;
        const float4 qn = make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
        const float d = g.x * qn.x + g.y * qn.y + g.z * qn.z + g.w * qn.w;
        o = make_float4((g.x - d * qn.x) * inv, (g.y - d * qn.y) * inv, (g.z - d * qn.z) * inv, (g.w - d * qn.w) * inv);
    } else {
        o = make_float4(g.x * 1e12f, g.y * 1e12f, g.z * 1e12f, g.w * 1e12f);
    }
    reinterpret_cast<float4*>(v_rotation_raw)[i] = o;
    const float sg = 1.f / (1.f + expf(-opacity_raw[i]));
    const float ds = sg * (1.f - sg);
    float go = v_opacities[i] * sg * (1.f - sg);
    if (opacity_reg != 0.f) go = fmaf(opacity_reg, ds, go);
    v_opacity_raw[i] = go;
}


template <int DEG> static int launch_sh_fwd(uint32_t n, uint32_t K, const float* dirs, const float* coeffs,
                                            const uint8_t* masks, float* colors, hipStream_t st) {
    constexpr uint32_t LS = ((DEG + 1) * (DEG + 1) * 3) | 1;
    const size_t lds = (size_t)SH_BLOCK * LS * sizeof(float);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(sh_fwd_kernel<DEG>), dim3((n + SH_BLOCK - 1) / SH_BLOCK), dim3(SH_BLOCK), lds, st,
                       n, K, dirs, coeffs, masks, colors);
    return 0;
}
template <int DEG> static int launch_sh_bwd(uint32_t n, uint32_t K, const float* dirs, const float* coeffs,
                                            const uint8_t* masks, const float* v_colors, float* v_coeffs, float* v_dirs,
                                            hipStream_t st) {
    const uint32_t LS = (K * 3u) | 1u;
    const size_t lds = (size_t)SH_BLOCK * LS * sizeof(float);
    const dim3 grid((n + SH_BLOCK - 1) / SH_BLOCK), block(SH_BLOCK);
    if (v_dirs)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(sh_bwd_kernel<DEG, true>), grid, block, lds, st, n, K, dirs, coeffs, masks,
                           v_colors, v_coeffs, v_dirs);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(sh_bwd_kernel<DEG, false>), grid, block, lds, st, n, K, dirs, coeffs, masks,
                           v_colors, v_coeffs, v_dirs);
    return 0;
}

int check_launch(const char* what);

}  // namespace gsx

using namespace gsx;

extern "C" int gsx_spherical_harmonics_fwd(uint32_t degrees_to_use, uint32_t n, uint32_t K, const float* dirs,
                                           const float* coeffs, const uint8_t* masks, float* colors, void* stream) {
    if (n == 0) return GSX_OK;  // upstream skips the launch (SphericalHarmonicsCUDA.cu:418-421)
    if (!dirs || !coeffs || !colors) { set_error("spherical_harmonics_fwd: null pointer"); return GSX_ERR_INVALID_ARGUMENT; }
    if (degrees_to_use > 4 || (degrees_to_use + 1) * (degrees_to_use + 1) > K) {
        set_error("spherical_harmonics_fwd: degrees_to_use needs (deg+1)^2 <= K and deg <= 4");
        return GSX_ERR_INVALID_ARGUMENT;
    }
    hipStream_t st = (hipStream_t)stream;
    switch (degrees_to_use) {
    case 0: launch_sh_fwd<0>(n, K, dirs, coeffs, masks, colors, st); break;
    case 1: launch_sh_fwd<1>(n, K, dirs, coeffs, masks, colors, st); break;
    case 2: launch_sh_fwd<2>(n, K, dirs, coeffs, masks, colors, st); break;
    case 3: launch_sh_fwd<3>(n, K, dirs, coeffs, masks, colors, st); break;
    default: launch_sh_fwd<4>(n, K, dirs, coeffs, masks, colors, st); break;
    }
    return check_launch("spherical_harmonics_fwd");
}

extern "C" int gsx_spherical_harmonics_bwd(uint32_t K, uint32_t degrees_to_use, uint32_t n, const float* dirs,
                                           const float* coeffs, const uint8_t* masks, const float* v_colors,
                                           float* v_coeffs, float* v_dirs, void* stream) {
    if (n == 0) return GSX_OK;
    if (!dirs || !coeffs || !v_colors || !v_coeffs) { set_error("spherical_harmonics_bwd: null pointer"); return GSX_ERR_INVALID_ARGUMENT; }
    if (degrees_to_use > 4 || (degrees_to_use + 1) * (degrees_to_use + 1) > K) {
        set_error("spherical_harmonics_bwd: degrees_to_use needs (deg+1)^2 <= K and deg <= 4");
        return GSX_ERR_INVALID_ARGUMENT;
    }
    if ((size_t)SH_BLOCK * ((K * 3u) | 1u) * sizeof(float) > 160u * 1024u) {
        set_error("spherical_harmonics_bwd: K too large for the LDS row tile (K <= 53)");
        return GSX_ERR_UNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    switch (degrees_to_use) {
    case 0: launch_sh_bwd<0>(n, K, dirs, coeffs, masks, v_colors, v_coeffs, v_dirs, st); break;
    case 1: launch_sh_bwd<1>(n, K, dirs, coeffs, masks, v_colors, v_coeffs, v_dirs, st); break;
    case 2: launch_sh_bwd<2>(n, K, dirs, coeffs, masks, v_colors, v_coeffs, v_dirs, st); break;
    case 3: launch_sh_bwd<3>(n, K, dirs, coeffs, masks, v_colors, v_coeffs, v_dirs, st); break;
    default: launch_sh_bwd<4>(n, K, dirs, coeffs, masks, v_colors, v_coeffs, v_dirs, st); break;
    }
    return check_launch("spherical_harmonics_bwd");
}

// ---- fused glue entry points (extensions; see include/gsx.h) ----------------------------------------------
extern "C" int gsx_sh_colors_fwd(uint32_t degrees_to_use, uint32_t C, uint32_t N, uint32_t K, const float* means,
                                 const float* viewmats, const float* coeffs, const int32_t* radii, float* colors, void* stream) {
    if (N == 0 || C == 0) return GSX_OK;
    if (!means || !viewmats || !coeffs || !radii || !colors) { set_error("sh_colors_fwd: null pointer"); return GSX_ERR_INVALID_ARGUMENT; }
    if (degrees_to_use > 4 || (degrees_to_use + 1) * (degrees_to_use + 1) > K) { set_error("sh_colors_fwd: bad degree"); return GSX_ERR_INVALID_ARGUMENT; }
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((N + SH_BLOCK - 1) / SH_BLOCK), block(SH_BLOCK);
    const uint32_t nq4 = (((degrees_to_use + 1) * (degrees_to_use + 1) * 3 + 3) / 4) * 4;
    // rows of whole 16 B vectors: the register variant (no LDS tile; 0.066 -> 0.062 ms in the S-1M step, 0.039 -> 0.034 ms cache-resident)
    const bool direct = (K * 3u) % 4u == 0u && nq4 <= K * 3u && (((uintptr_t)coeffs) & 15u) == 0;
    if (direct) {
#define GSX_L(D) hipLaunchKernelGGL(HIP_KERNEL_NAME(sh_colors_fwd_direct_kernel<D>), grid, block, 0, st, C, N, K, means, viewmats, coeffs, radii, colors)
        switch (degrees_to_use) { case 0: GSX_L(0); break; case 1: GSX_L(1); break; case 2: GSX_L(2); break; case 3: GSX_L(3); break; default: GSX_L(4); break; }
#undef GSX_L
        return check_launch("sh_colors_fwd");
    }
#define GSX_L(D) hipLaunchKernelGGL(HIP_KERNEL_NAME(sh_colors_fwd_kernel<D>), grid, block, (size_t)SH_BLOCK * ((((D + 1) * (D + 1) * 3) | 1)) * 4, st, C, N, K, means, viewmats, coeffs, radii, colors)
    switch (degrees_to_use) { case 0: GSX_L(0); break; case 1: GSX_L(1); break; case 2: GSX_L(2); break; case 3: GSX_L(3); break; default: GSX_L(4); break; }
#undef GSX_L
    return check_launch("sh_colors_fwd");
}

extern "C" int gsx_sh_colors_bwd(uint32_t degrees_to_use, uint32_t C, uint32_t N, uint32_t K, const float* means,
                                 const float* viewmats, const float* coeffs, const int32_t* radii, const float* colors,
                                 const float* v_colors, float* v_coeffs, const float* v_means_in, float* v_means_out, void* stream) {
    if (N == 0 || C == 0) return GSX_OK;
    if (!means || !viewmats || !coeffs || !v_colors || !v_coeffs || !v_means_out) { set_error("sh_colors_bwd: null pointer"); return GSX_ERR_INVALID_ARGUMENT; }
    if ((radii == nullptr) != (colors == nullptr)) { set_error("sh_colors_bwd: radii and colors are given together (or neither: pre-masked v_colors)"); return GSX_ERR_INVALID_ARGUMENT; }
    if (degrees_to_use > 4 || (degrees_to_use + 1) * (degrees_to_use + 1) > K) { set_error("sh_colors_bwd: bad degree"); return GSX_ERR_INVALID_ARGUMENT; }
    const size_t lds = (size_t)SH_BLOCK * ((K * 3u) | 1u) * sizeof(float);
    if (lds > 160u * 1024u) { set_error("sh_colors_bwd: K too large for the LDS row tile (K <= 53)"); return GSX_ERR_UNSUPPORTED; }
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((N + SH_BLOCK - 1) / SH_BLOCK), block(SH_BLOCK);
#define GSX_L(D) hipLaunchKernelGGL(HIP_KERNEL_NAME(sh_colors_bwd_kernel<D, false>), grid, block, lds, st, C, N, K, means, viewmats, coeffs, radii, colors, v_colors, v_coeffs, v_means_in, v_means_out, ShAdam{})
    switch (degrees_to_use) { case 0: GSX_L(0); break; case 1: GSX_L(1); break; case 2: GSX_L(2); break; case 3: GSX_L(3); break; default: GSX_L(4); break; }
#undef GSX_L
    return check_launch("sh_colors_bwd");
}

// gsx_sh_colors_bwd fused with the Adam step of the SH tensor (the sh0 / shN groups of src/training/optimizers/fused_adam.cpp:20-96, same
// arithmetic as gsx_adam_step_split): coeffs [N,K,3], exp_avg, exp_avg_sq are updated in place, the SH gradient is never materialised.
// K * 3 must be a multiple of 4 (K = 4, 16: degrees 1 and 3); step_* = lr * bias_correction1_rcp; do_* = 0 leaves that block untouched.
extern "C" int gsx_sh_colors_bwd_adam(uint32_t degrees_to_use, uint32_t C, uint32_t N, uint32_t K, const float* means,
                                      const float* viewmats, float* coeffs, const int32_t* radii, const float* colors,
                                      const float* v_colors, const float* v_means_in, float* v_means_out, float* exp_avg,
                                      float* exp_avg_sq, float step_sh0, float step_shN, int do_sh0, int do_shN, float beta1, float beta2,
                                      float eps, float bias_correction2_sqrt_rcp, void* stream) {
    if (N == 0 || C == 0) return GSX_OK;
    if (!means || !viewmats || !coeffs || !v_colors || !v_means_out || !exp_avg || !exp_avg_sq) { set_error("sh_colors_bwd_adam: null pointer"); return GSX_ERR_INVALID_ARGUMENT; }
    if ((radii == nullptr) != (colors == nullptr)) { set_error("sh_colors_bwd_adam: radii and colors are given together (or neither: pre-masked v_colors)"); return GSX_ERR_INVALID_ARGUMENT; }
    if (degrees_to_use > 4 || (degrees_to_use + 1) * (degrees_to_use + 1) > K) { set_error("sh_colors_bwd_adam: bad degree"); return GSX_ERR_INVALID_ARGUMENT; }
    if ((K * 3u) % 4u != 0u || ((((uintptr_t)coeffs | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15u) != 0)) {
        set_error("sh_colors_bwd_adam: K * 3 must be a multiple of 4 and the arrays 16-byte aligned");
        return GSX_ERR_UNSUPPORTED;
    }
    const size_t lds = (size_t)SH_BLOCK * ((K * 3u) | 1u) * sizeof(float);
    if (lds > 160u * 1024u) { set_error("sh_colors_bwd_adam: K too large for the LDS row tile (K <= 53)"); return GSX_ERR_UNSUPPORTED; }
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((N + SH_BLOCK - 1) / SH_BLOCK), block(SH_BLOCK);
    const ShAdam ad{exp_avg, exp_avg_sq, step_sh0, step_shN, do_sh0, do_shN, beta1, beta2, eps, bias_correction2_sqrt_rcp};
#define GSX_L(D) hipLaunchKernelGGL(HIP_KERNEL_NAME(sh_colors_bwd_kernel<D, true>), grid, block, lds, st, C, N, K, means, viewmats, coeffs, radii, colors, v_colors, coeffs, v_means_in, v_means_out, ad)
    switch (degrees_to_use) { case 0: GSX_L(0); break; case 1: GSX_L(1); break; case 2: GSX_L(2); break; case 3: GSX_L(3); break; default: GSX_L(4); break; }
#undef GSX_L
    return check_launch("sh_colors_bwd_adam");
}

extern "C" int gsx_splat_activations_fwd(uint32_t N, const float* scaling_raw, const float* rotation_raw, const float* opacity_raw,
                                         float* scales, float* quats, float* opacities, void* stream) {
    if (N == 0) return GSX_OK;
    if (!scaling_raw || !rotation_raw || !opacity_raw || !scales || !quats || !opacities) { set_error("splat_activations_fwd: null pointer"); return GSX_ERR_INVALID_ARGUMENT; }
    hipLaunchKernelGGL(splat_activations_fwd_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, scaling_raw, rotation_raw,
                       opacity_raw, scales, quats, opacities);
    return check_launch("splat_activations_fwd");
}

extern "C" int gsx_splat_activations_bwd(uint32_t N, const float* scaling_raw, const float* rotation_raw, const float* opacity_raw,
                                         const float* v_scales, const float* v_quats, const float* v_opacities, float* v_scaling_raw,
                                         float* v_rotation_raw, float* v_opacity_raw, void* stream) {
    return gsx_splat_activations_bwd_reg(N, scaling_raw, rotation_raw, opacity_raw, v_scales, v_quats, v_opacities, v_scaling_raw, v_rotation_raw,
                                         v_opacity_raw, 0.f, 0.f, stream);
}

extern "C" int gsx_splat_activations_bwd_reg(uint32_t N, const float* scaling_raw, const float* rotation_raw, const float* opacity_raw,
                                             const float* v_scales, const float* v_quats, const float* v_opacities, float* v_scaling_raw,
                                             float* v_rotation_raw, float* v_opacity_raw, float scale_reg_per_element, float opacity_reg_per_element,
                                             void* stream) {
    if (N == 0) return GSX_OK;
    if (!scaling_raw || !rotation_raw || !opacity_raw || !v_scales || !v_quats || !v_opacities || !v_scaling_raw || !v_rotation_raw || !v_opacity_raw) {
        set_error("splat_activations_bwd: null pointer");
        return GSX_ERR_INVALID_ARGUMENT;
    }
    hipLaunchKernelGGL(splat_activations_bwd_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, scaling_raw, rotation_raw,
                       opacity_raw, v_scales, v_quats, v_opacities, v_scaling_raw, v_rotation_raw, v_opacity_raw, scale_reg_per_element, opacity_reg_per_element);
    return check_launch("splat_activations_bwd");
}
