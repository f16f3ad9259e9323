// gsx_ssim.hip — photometric loss of the training step: the step right after the blend, producing
// v_render_colors for the backward.
//   fused SSIM fwd / bwd   src/training/kernels/ssim.cu:64-275 / 283-428 (host wrappers :436-510),
//                          autograd wrapper include/kernels/fused_ssim.cuh:26-122 ("valid" = crop 5 px of the map)
//   loss composition       src/training/trainer.cpp:103-127: (1-lambda) * l1_loss + lambda * (1 - mean(ssim_valid))
//   image clamp            src/training/rasterization/rasterizer.cpp:401 (clamp(render, 0, 1) before the loss)
//
// MI355X design.  An 11x11 separable Gaussian window over 5 (fwd) / 3 (bwd) per-pixel quantities; the work is LDS-bound,
// not HBM-bound, so the kernels minimise LDS traffic instead of replaying the usual one-output-per-thread stencil:
//   * a block of 512 threads owns a 64x32 pixel tile (one wave64 = one 64-pixel row segment -> 256 B coalesced row reads / writes; the
//     10-pixel halo costs 1.52x the tile in fetches and 1.31x in horizontal-pass work — 1.88x / 1.63x with the 64x16 tile of round 2);
//   * the horizontal pass gives every thread a strip of 4 outputs in one row (14+14 LDS reads for 4x5 outputs instead
//     of 22 per output) with lanes running down the rows — odd LDS pitches (75 / 65 dwords) keep both the strip reads
//     and the plane writes bank-conflict free;
//   * the vertical pass gives every thread a strip of 4 outputs in one column (14 reads per quantity for 4 outputs),
//     lanes along x, conflict-free by construction;
//   * the fused loss kernels read the blend's [H,W,3] output directly (clamp on the fly), pre-multiply the three SSIM
//     partial-derivative maps with the (constant) upstream gradient, reduce the two loss sums per block, and the backward
//     writes v_render_colors in the blend's own [H,W,3] layout with the L1 term and the clamp mask folded in: the whole
//     loss is two tile kernels and a 1-block finaliser, no permute / clamp / crop / mean / l1 kernels in between.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gsx.h"

namespace gsx {

void set_error(const char* msg);
int check_launch(const char* what);

namespace ssim {

constexpr int TX = 64, TY = 32, HALO = 5;
constexpr int SX = TX + 2 * HALO, SY = TY + 2 * HALO;  // staged tile 74 x 42
constexpr int PA = 75;                                  // pitch of the staged input planes (odd: rows -> distinct banks)
constexpr int PC = 65;                                  // pitch of the horizontally-convolved planes
// (round 4: pitches = 3 mod 16 (83 / 67) remove the remaining 2-way conflict between the two row groups of a wave in the horizontal
// pass — measured flat, 0.0543 / 0.0443 ms either way: the kernels are bound by their phase structure, not by LDS conflicts)
constexpr int NT = 512;                                 // 8 waves: wave w owns rows 4w .. 4w+3 of the tile in the vertical pass

// the reference's window: normalised Gaussian, sigma 1.5, 11 taps, as fp32 literals (ssim.cu:16-27)
#define GSX_SSIM_TAPS                                                                                                          \
    {0.001028380123898387f, 0.0075987582094967365f, 0.036000773310661316f, 0.10936068743467331f, 0.21300552785396576f,        \
     0.26601171493530273f, 0.21300552785396576f, 0.10936068743467331f, 0.036000773310661316f, 0.0075987582094967365f,         \
     0.001028380123898387f}

// horizontal pass, 5 quantities (X, X^2, Y, Y^2, XY) from two staged planes
__device__ __forceinline__ void hconv5(const float (*sA)[PA], const float (*sB)[PA], float (*sC)[SY][PC]) {
    constexpr float w[11] = GSX_SSIM_TAPS;
    for (int it = threadIdx.x; it < SY * (TX / 4); it += NT) {
        const int r = it % SY, x0 = (it / SY) * 4;
        float X[14], Y[14], XX[14], YY[14], XY[14];   // the three products once per staged element (42 mul per 4 outputs) instead of per tap
#pragma unroll
        for (int k = 0; k < 14; ++k) {
            X[k] = sA[r][x0 + k];
            Y[k] = sB[r][x0 + k];
            XX[k] = X[k] * X[k]; YY[k] = Y[k] * Y[k]; XY[k] = X[k] * Y[k];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                a0 = fmaf(w[k], X[j + k], a0);
                a1 = fmaf(w[k], XX[j + k], a1);
                a2 = fmaf(w[k], Y[j + k], a2);
                a3 = fmaf(w[k], YY[j + k], a3);
                a4 = fmaf(w[k], XY[j + k], a4);
            }
            sC[0][r][x0 + j] = a0;
            sC[1][r][x0 + j] = a1;
            sC[2][r][x0 + j] = a2;
            sC[3][r][x0 + j] = a3;
            sC[4][r][x0 + j] = a4;
        }
    }
}

// horizontal pass of the fused training loss: FOUR quantities (X, Y, X^2 + Y^2, XY).  SSIM and its three partials depend on the two
// windowed variances only through their SUM (ssim.cu:255-268: B = sigma1^2 + sigma2^2 + C2 is the only place they enter), and the window
// is linear, so one convolution of X^2 + Y^2 replaces the two of X^2 and Y^2: a fifth fewer taps in both passes and a fifth less LDS.
__device__ __forceinline__ void hconv4(const float (*sA)[PA], const float (*sB)[PA], float (*sC)[SY][PC]) {
    constexpr float w[11] = GSX_SSIM_TAPS;
    for (int it = threadIdx.x; it < SY * (TX / 4); it += NT) {
        const int r = it % SY, x0 = (it / SY) * 4;
        float X[14], Y[14], SQ[14], XY[14];
#pragma unroll
        for (int k = 0; k < 14; ++k) {
            X[k] = sA[r][x0 + k];
            Y[k] = sB[r][x0 + k];
            SQ[k] = fmaf(X[k], X[k], Y[k] * Y[k]); XY[k] = X[k] * Y[k];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                a0 = fmaf(w[k], X[j + k], a0);
                a1 = fmaf(w[k], Y[j + k], a1);
                a2 = fmaf(w[k], SQ[j + k], a2);
                a3 = fmaf(w[k], XY[j + k], a3);
            }
            sC[0][r][x0 + j] = a0;
            sC[1][r][x0 + j] = a1;
            sC[2][r][x0 + j] = a2;
            sC[3][r][x0 + j] = a3;
        }
    }
}

// horizontal pass, 3 already-formed quantities
__device__ __forceinline__ void hconv3(const float (*sD)[SY][PA], float (*sC)[SY][PC]) {
    constexpr float w[11] = GSX_SSIM_TAPS;
    for (int it = threadIdx.x; it < SY * (TX / 4); it += NT) {
        const int r = it % SY, x0 = (it / SY) * 4;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            float V[14];
#pragma unroll
            for (int k = 0; k < 14; ++k) V[k] = sD[q][r][x0 + k];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a = 0.f;
#pragma unroll
                for (int k = 0; k < 11; ++k) a = fmaf(w[k], V[j + k], a);
                sC[q][r][x0 + j] = a;
            }
        }
    }
}

// vertical pass: thread (x, wave) -> rows wave*4 .. wave*4+3
template <int NQ>
__device__ __forceinline__ void vconv(const float (*sC)[SY][PC], int x, int y0, float (&out)[4][NQ]) {
    constexpr float w[11] = GSX_SSIM_TAPS;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        float V[14];
#pragma unroll
        for (int k = 0; k < 14; ++k) V[k] = sC[q][y0 + k][x];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < 11; ++k) a = fmaf(w[k], V[j + k], a);
            out[j][q] = a;
        }
    }
}

// SSIM value and its partials w.r.t. mu1, sigma1^2, sigma12 from the five windowed moments (ssim.cu:240-270)
__device__ __forceinline__ void ssim_point(const float (&o)[5], float C1, float C2, float& val, float& d_mu1, float& d_s1, float& d_s12) {
    const float mu1 = o[0], mu2 = o[2];
    const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2;
    const float s1 = o[1] - mu1_sq, s2 = o[3] - mu2_sq, s12 = o[4] - mu1 * mu2;
    const float A = mu1_sq + mu2_sq + C1, B = s1 + s2 + C2;
    const float Cn = 2.f * mu1 * mu2 + C1, Dn = 2.f * s12 + C2;
    // the reference's expressions (ssim.cu:255-268) with 1/A and 1/B formed once: two divisions instead of four
    const float rA = 1.f / A, rB = 1.f / B, rAB = rA * rB;
    val = Cn * Dn * rAB;
    const float t = (mu1 * 2.f) * val;          // mu1 2 Cn Dn / (A B)
    d_mu1 = (mu2 * 2.f) * (Dn - Cn) * rAB - t * rA + t * rB;
    d_s1 = -val * rB;
    d_s12 = (2.f * Cn) * rAB;
}

// the same from the four moments of hconv4: o = (mu1, mu2, E[x^2 + y^2], E[xy])
__device__ __forceinline__ void ssim_point4(const float (&o)[4], float C1, float C2, float& val, float& d_mu1, float& d_s1, float& d_s12) {
    const float mu1 = o[0], mu2 = o[1];
    const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2;
    const float s_sum = (o[2] - mu1_sq) - mu2_sq, s12 = o[3] - mu1 * mu2;   // sigma1^2 + sigma2^2, sigma12
    const float A = mu1_sq + mu2_sq + C1, B = s_sum + C2;
    const float Cn = 2.f * mu1 * mu2 + C1, Dn = 2.f * s12 + C2;
    const float rA = 1.f / A, rB = 1.f / B, rAB = rA * rB;
    val = Cn * Dn * rAB;
    const float t = (mu1 * 2.f) * val;
    d_mu1 = (mu2 * 2.f) * (Dn - Cn) * rAB - t * rA + t * rB;
    d_s1 = -val * rB;
    d_s12 = (2.f * Cn) * rAB;
}

// ---- the reference's operator pair on planar [B,CH,H,W] images ------------------------------------------------------------
template <bool TRAIN>
__global__ __launch_bounds__(NT) void ssim_fwd_kernel(int CH, int H, int W, float C1, float C2, const float* __restrict__ img1,
                                                      const float* __restrict__ img2, float* __restrict__ ssim_map,
                                                      float* __restrict__ dm_dmu1, float* __restrict__ dm_dsigma1_sq,
                                                      float* __restrict__ dm_dsigma12) {
    __shared__ float sA[SY][PA], sB[SY][PA];
    __shared__ float sC[5][SY][PC];
    const int tx0 = blockIdx.x * TX, ty0 = blockIdx.y * TY;
    const int x = threadIdx.x & 63, y0 = (threadIdx.x >> 6) * 4;
    const size_t npix = (size_t)H * W;
    for (int c = 0; c < CH; ++c) {
        const size_t plane = ((size_t)blockIdx.z * CH + c) * npix;
        for (int i = threadIdx.x; i < SY * SX; i += NT) {
            const int ly = i / SX, lx = i - ly * SX;
            const int gy = ty0 + ly - HALO, gx = tx0 + lx - HALO;
            const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;
            sA[ly][lx] = in ? img1[plane + (size_t)gy * W + gx] : 0.f;
            sB[ly][lx] = in ? img2[plane + (size_t)gy * W + gx] : 0.f;
        }
        __syncthreads();
        hconv5(sA, sB, sC);
        __syncthreads();
        float o[4][5];
        vconv<5>(sC, x, y0, o);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gx = tx0 + x, gy = ty0 + y0 + j;
            if (gx < W && gy < H) {
                float val, d0, d1, d2;
                ssim_point(o[j], C1, C2, val, d0, d1, d2);
                const size_t idx = plane + (size_t)gy * W + gx;
                ssim_map[idx] = val;
                if (TRAIN) {
                    dm_dmu1[idx] = d0;
                    dm_dsigma1_sq[idx] = d1;
                    dm_dsigma12[idx] = d2;
                }
            }
        }
    }
}

__global__ __launch_bounds__(NT) void ssim_bwd_kernel(int CH, int H, int W, const float* __restrict__ img1, const float* __restrict__ img2,
                                                      const float* __restrict__ dL_dmap, float* __restrict__ dL_dimg1,
                                                      const float* __restrict__ dm_dmu1, const float* __restrict__ dm_dsigma1_sq,
                                                      const float* __restrict__ dm_dsigma12) {
    __shared__ float sD[3][SY][PA];
    __shared__ float sC[3][SY][PC];
    const int tx0 = blockIdx.x * TX, ty0 = blockIdx.y * TY;
    const int x = threadIdx.x & 63, y0 = (threadIdx.x >> 6) * 4;
    const size_t npix = (size_t)H * W;
    for (int c = 0; c < CH; ++c) {
        const size_t plane = ((size_t)blockIdx.z * CH + c) * npix;
        for (int i = threadIdx.x; i < SY * SX; i += NT) {
            const int ly = i / SX, lx = i - ly * SX;
            const int gy = ty0 + ly - HALO, gx = tx0 + lx - HALO;
            float a = 0.f, b = 0.f, d = 0.f;
            if (gx >= 0 && gx < W && gy >= 0 && gy < H) {
                const size_t idx = plane + (size_t)gy * W + gx;
                const float chain = dL_dmap[idx];
                a = dm_dmu1[idx] * chain;
                b = dm_dsigma1_sq[idx] * chain;
                d = dm_dsigma12[idx] * chain;
            }
            sD[0][ly][lx] = a;
            sD[1][ly][lx] = b;
            sD[2][ly][lx] = d;
        }
        __syncthreads();
        hconv3(sD, sC);
        __syncthreads();
        float s[4][3];
        vconv<3>(sC, x, y0, s);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gx = tx0 + x, gy = ty0 + y0 + j;
            if (gx < W && gy < H) {
                const size_t idx = plane + (size_t)gy * W + gx;
                dL_dimg1[idx] = s[j][0] + (2.f * img1[idx]) * s[j][1] + img2[idx] * s[j][2];
            }
        }
    }
}

__host__ __device__ __forceinline__ uint32_t loss_blocks_per_image(int H, int W) { return (uint32_t)((W + TX - 1) / TX) * (uint32_t)((H + TY - 1) / TY); }
__device__ __forceinline__ bool loss_tile(int W, int H, int& tx0, int& ty0, int& tile_id) {
    const uint32_t ntx = (uint32_t)(W + TX - 1) / TX, n = ntx * ((uint32_t)(H + TY - 1) / TY), per = (n + 7u) / 8u, b = blockIdx.x;
    const uint32_t t = (b & 7u) * per + (b >> 3);
    if ((b >> 3) >= per || t >= n) return false;
    tile_id = (int)t;
    ty0 = (int)(t / ntx) * TY; tx0 = (int)(t % ntx) * TX;
    return true;
}

// ---- fused photometric loss on the blend's own layout ----------------------------------------------------------------------
// render [C,H,W,3] (unclamped blend output), gt [C,3,H,W].  Workspace: chained partial maps [3][C][3][H][W] + block partial sums.
__global__ __launch_bounds__(NT, 4) void loss_fwd_kernel(int H, int W, float chain, int crop, const float* __restrict__ render,
                                                      const float* __restrict__ gt, float* __restrict__ maps,
                                                      float2* __restrict__ block_sums) {
    __shared__ float sA[SY][PA], sB[SY][PA];
    __shared__ float sC[4][SY][PC];
    __shared__ float2 s_red[NT / 64];
    int tx0, ty0, tile_id;
    if (!loss_tile(W, H, tx0, ty0, tile_id)) return;
    const int x = threadIdx.x & 63, y0 = (threadIdx.x >> 6) * 4;
    const size_t npix = (size_t)H * W;
    const size_t q_stride = (size_t)gridDim.z * 3 * npix;
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    float l1 = 0.f, ss = 0.f;
    // (Round 4 tried to software-pipeline the staging — the global loads of channel c + 1 issued right after the horizontal pass of channel c,
    // held in 14 registers across the vertical pass: the kernel sits at its 128-VGPR cap (4 waves / SIMD for two 512-thread blocks per CU),
    // the prefetch registers spilled (92 B of scratch per lane) and the forward went from 0.063 to 0.074 ms.  Not kept.)
    for (int c = 0; c < 3; ++c) {
        const size_t plane = ((size_t)blockIdx.z * 3 + c) * npix;
        const float* rbase = render + (size_t)blockIdx.z * npix * 3 + c;
        for (int i = threadIdx.x; i < SY * SX; i += NT) {
            const int ly = i / SX, lx = i - ly * SX;
            const int gy = ty0 + ly - HALO, gx = tx0 + lx - HALO;
            const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;
            const size_t p = (size_t)gy * W + gx;
            sA[ly][lx] = in ? fminf(fmaxf(rbase[p * 3], 0.f), 1.f) : 0.f;
            sB[ly][lx] = in ? gt[plane + p] : 0.f;
        }
        __syncthreads();
        float adiff[4];  // |X - Y| of this thread's pixels, read before the planes are restaged for the next channel
#pragma unroll
        for (int j = 0; j < 4; ++j) adiff[j] = fabsf(sA[y0 + j + HALO][x + HALO] - sB[y0 + j + HALO][x + HALO]);
        hconv4(sA, sB, sC);
        __syncthreads();
        float o[4][4];
        vconv<4>(sC, x, y0, o);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gx = tx0 + x, gy = ty0 + y0 + j;
            if (gx < W && gy < H) {
                float val, d0, d1, d2;
                ssim_point4(o[j], C1, C2, val, d0, d1, d2);
                const bool valid = gx >= crop && gx < W - crop && gy >= crop && gy < H - crop;
                const float ch = valid ? chain : 0.f;
                const size_t idx = plane + (size_t)gy * W + gx;
                // (ordinary stores: the backward runs right behind and finds the maps in the Infinity Cache; stored nontemporally the forward
                // gains 3 us — the maps no longer evict the render / target lines the next channel pass re-reads — and the backward loses 11)
                maps[idx] = d0 * ch;
                maps[q_stride + idx] = d1 * ch;
                maps[2 * q_stride + idx] = d2 * ch;
                ss += valid ? val : 0.f;
                l1 += adiff[j];
            }
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        l1 += __shfl_xor(l1, m);
        ss += __shfl_xor(ss, m);
    }
    if (x == 0) s_red[threadIdx.x >> 6] = make_float2(l1, ss);
    __syncthreads();
    if (threadIdx.x == 0) {
        float2 t = s_red[0];
#pragma unroll
        for (int k = 1; k < NT / 64; ++k) {
            t.x += s_red[k].x;
            t.y += s_red[k].y;
        }
        block_sums[(size_t)blockIdx.z * loss_blocks_per_image(H, W) + (size_t)tile_id] = t;
    }
}

// loss3 = {loss, l1 mean, ssim mean}
__global__ __launch_bounds__(NT) void loss_finalize_kernel(uint32_t n_blocks, const float2* __restrict__ block_sums, double inv_l1,
                                                           double inv_ssim, float lambda, float* __restrict__ loss3) {
    __shared__ double s0[NT], s1[NT];
    double a = 0.0, b = 0.0;
    for (uint32_t i = threadIdx.x; i < n_blocks; i += NT) {
        a += (double)block_sums[i].x;
        b += (double)block_sums[i].y;
    }
    s0[threadIdx.x] = a;
    s1[threadIdx.x] = b;
    __syncthreads();
    for (int m = NT / 2; m >= 1; m >>= 1) {
        if ((int)threadIdx.x < m) {
            s0[threadIdx.x] += s0[threadIdx.x + m];
            s1[threadIdx.x] += s1[threadIdx.x + m];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double l1 = s0[0] * inv_l1, ssim = s1[0] * inv_ssim;
        loss3[0] = (float)((1.0 - (double)lambda) * l1 + (double)lambda * (1.0 - ssim));
        loss3[1] = (float)l1;
        loss3[2] = (float)ssim;
    }
}

__global__ __launch_bounds__(NT) void loss_bwd_kernel(int H, int W, float l1_coeff, const float* __restrict__ grad_loss, float grad_scale,
                                                      const float* __restrict__ render, const float* __restrict__ gt,
                                                      const float* __restrict__ maps, float* __restrict__ v_render) {
    __shared__ float sD[3][SY][PA];
    __shared__ float sC[3][SY][PC];
    int tx0, ty0, tile_id;
    if (!loss_tile(W, H, tx0, ty0, tile_id)) return;
    const int x = threadIdx.x & 63, y0 = (threadIdx.x >> 6) * 4;
    const size_t npix = (size_t)H * W;
    const size_t q_stride = (size_t)gridDim.z * 3 * npix;
    const float up = grad_scale * (grad_loss ? grad_loss[0] : 1.f);
    float g[4][3];
    // Staging is software-pipelined over the three channels: the loads of channel c + 1's three maps are issued right after the horizontal
    // pass of channel c (which frees sD) and land in 21 registers while the vertical pass and the output arithmetic of channel c run (the
    // kernel has the room: 74 VGPRs of 128; the forward, at its cap, does not — see loss_fwd_kernel).
    constexpr int NS = (SY * SX + NT - 1) / NT;   // staged elements per thread and map (7)
    float pm[3][NS];
    auto fetch = [&](int c) {
        const size_t plane = ((size_t)blockIdx.z * 3 + c) * npix;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int i = threadIdx.x + k * NT;
            const int ly = i / SX, lx = i - ly * SX;
            const int gy = ty0 + ly - HALO, gx = tx0 + lx - HALO;
            const bool in = i < SY * SX && gx >= 0 && gx < W && gy >= 0 && gy < H;
            const size_t idx = plane + (size_t)gy * W + gx;
            pm[0][k] = in ? maps[idx] : 0.f;   // (plain loads: neighbouring tiles share the halo lines through L2; nontemporal loads: 0.044 -> 0.055 ms)
            pm[1][k] = in ? maps[q_stride + idx] : 0.f;
            pm[2][k] = in ? maps[2 * q_stride + idx] : 0.f;
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int i = threadIdx.x + k * NT;
            if (i < SY * SX) {
                const int ly = i / SX, lx = i - ly * SX;
                sD[0][ly][lx] = pm[0][k]; sD[1][ly][lx] = pm[1][k]; sD[2][ly][lx] = pm[2][k];
            }
        }
    };
    fetch(0);
    stash();
    __syncthreads();
    for (int c = 0; c < 3; ++c) {
        const size_t plane = ((size_t)blockIdx.z * 3 + c) * npix;
        hconv3(sD, sC);
        __syncthreads();
        if (c < 2) fetch(c + 1);   // in flight during the vertical pass
        float s[4][3];
        vconv<3>(sC, x, y0, s);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gx = tx0 + x, gy = ty0 + y0 + j;
            float v = 0.f;
            if (gx < W && gy < H) {
                const size_t p = (size_t)gy * W + gx;
                const float raw = render[((size_t)blockIdx.z * npix + p) * 3 + c];
                const float p1 = fminf(fmaxf(raw, 0.f), 1.f), p2 = gt[plane + p];
                const float d = p1 - p2;
                const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
                v = s[j][0] + (2.f * p1) * s[j][1] + p2 * s[j][2] + l1_coeff * sgn;
                v = (raw >= 0.f && raw <= 1.f) ? v * up : 0.f;  // clamp backward (inclusive bounds, as torch)
            }
            g[j][c] = v;
        }
        if (c < 2) {
            stash();           // sD is free since the barrier above; sC is still being read by slower waves: the next hconv3 waits below
            __syncthreads();
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int gx = tx0 + x, gy = ty0 + y0 + j;
        if (gx < W && gy < H) {
            float* o = v_render + ((size_t)blockIdx.z * npix + (size_t)gy * W + gx) * 3;
            o[0] = g[j][0];
            o[1] = g[j][1];
            o[2] = g[j][2];
        }
    }
}

// ---- single-pass training loss + gradient (round 6; VERDICT r05 next #2a: built to be MEASURED against the pair above) ------------------------------
// One kernel: the SSIM statistics are recomputed on the output tile + a 5-pixel ring, the three chained derivative maps live in LDS only, and their
// 11 x 11 windows produce v_render for the tile — no 75 MB of maps written and re-read, one staging instead of two.  The price is the ring: on a 32 x 32
// tile (two 256-thread blocks per CU by LDS: 57.8 KB) staging covers 52 x 52 (2.64 x the tile), the first horizontal pass 52 x 42, the first vertical pass
// 42 x 42, the second horizontal pass 42 x 32: 22.3 convolution units per pixel against 16.2 for the pair.  Same arithmetic per output (same tap order):
// v_render and the loss sums equal the pair's (tests/test_loss.py / test_gpu_reference_train.py compare).  `up` = the constant upstream gradient.
namespace sp {
constexpr int T = 32, R1 = T + 2 * HALO, R2 = T + 4 * HALO;   // tile 32, map region 42, staged region 52
constexpr int PS = R2 + 1, PH = R1 + 1, PE = T + 1;          // odd pitches: 53 (staged), 43 (first-pass planes, maps), 33 (second-pass planes)
constexpr int NTS = 256;
}
__global__ __launch_bounds__(sp::NTS, 2) void loss_single_pass_kernel(int H, int W, float chain, int crop, float l1_coeff, float up, const float* __restrict__ render,
                                                                       const float* __restrict__ gt, float* __restrict__ v_render, float2* __restrict__ block_sums) {
    using namespace sp;
    constexpr float w[11] = GSX_SSIM_TAPS;
    __shared__ float s_in[2 * R2 * PS];          // X, Y over the staged region; later the three maps [3][R1][PH]
    __shared__ float s_cv[4 * R2 * PH];          // first horizontal pass [4][R2][PH]; later the second horizontal pass [3][R1][PE]
    __shared__ float2 s_red[NTS / 64];
    static_assert(3 * R1 * PH <= 2 * R2 * PS && 3 * R1 * PE <= 4 * R2 * PH, "aliases");
    float (*sA)[PS] = reinterpret_cast<float (*)[PS]>(s_in), (*sB)[PS] = reinterpret_cast<float (*)[PS]>(s_in + R2 * PS);
    float (*sC)[R2][PH] = reinterpret_cast<float (*)[R2][PH]>(s_cv);
    float (*sD)[R1][PH] = reinterpret_cast<float (*)[R1][PH]>(s_in);
    float (*sE)[R1][PE] = reinterpret_cast<float (*)[R1][PE]>(s_cv);
    const uint32_t ntx = (uint32_t)(W + T - 1) / T, nty = (uint32_t)(H + T - 1) / T, n = ntx * nty, per = (n + 7u) / 8u, b = blockIdx.x;
    const uint32_t tix = (b & 7u) * per + (b >> 3);   // XCD-aware: every XCD a contiguous row-major run of tiles (loss_tile)
    if ((b >> 3) >= per || tix >= n) return;
    const int tx0 = (int)(tix % ntx) * T, ty0 = (int)(tix / ntx) * T;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ox = tid & 31, oy0 = (tid >> 5) * 4;    // this thread's four output pixels: column ox, rows oy0 .. oy0 + 3
    const size_t npix = (size_t)H * W;
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    float l1 = 0.f, ss = 0.f, g[4][3];
    for (int c = 0; c < 3; ++c) {
        const size_t plane = ((size_t)blockIdx.z * 3 + c) * npix;
        const float* rbase = render + (size_t)blockIdx.z * npix * 3 + c;
        // ---- stage X (clamped render) and Y over 52 x 52: wave -> rows, lane -> column
        for (int r = wave; r < R2; r += NTS / 64) {
            const int gy = ty0 + r - 2 * HALO;
            if (lane < R2) {
                const int gx = tx0 + lane - 2 * HALO;
                const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;
                const size_t p = (size_t)gy * W + gx;
                sA[r][lane] = in ? fminf(fmaxf(rbase[p * 3], 0.f), 1.f) : 0.f;
                sB[r][lane] = in ? gt[plane + p] : 0.f;
            }
        }
        __syncthreads();
        float p1[4], p2[4];   // X, Y of this thread's output pixels (the staged planes are overwritten by the maps below)
#pragma unroll
        for (int j = 0; j < 4; ++j) { p1[j] = sA[oy0 + j + 2 * HALO][ox + 2 * HALO]; p2[j] = sB[oy0 + j + 2 * HALO][ox + 2 * HALO]; }
        // ---- first horizontal pass: 52 rows x 42 columns, strips of 3 outputs (14 per row: 728 items on 256 threads)
        for (int it = tid; it < R2 * (R1 / 3); it += NTS) {
            const int r = it % R2, x0 = (it / R2) * 3;
            float X[13], Y[13], SQ[13], XY[13];
#pragma unroll
            for (int k = 0; k < 13; ++k) {
                X[k] = sA[r][x0 + k]; Y[k] = sB[r][x0 + k];
                SQ[k] = fmaf(X[k], X[k], Y[k] * Y[k]); XY[k] = X[k] * Y[k];
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
                for (int k = 0; k < 11; ++k) {
                    a0 = fmaf(w[k], X[j + k], a0); a1 = fmaf(w[k], Y[j + k], a1);
                    a2 = fmaf(w[k], SQ[j + k], a2); a3 = fmaf(w[k], XY[j + k], a3);
                }
                sC[0][r][x0 + j] = a0; sC[1][r][x0 + j] = a1; sC[2][r][x0 + j] = a2; sC[3][r][x0 + j] = a3;
            }
        }
        __syncthreads();
        // ---- first vertical pass + SSIM point + chained partials: 42 columns x 6 strips of 7 rows (252 items)
        if (tid < R1 * 6) {
            const int x = tid % R1, y0 = (tid / R1) * 7;
            float o[7][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float V[17];
#pragma unroll
                for (int k = 0; k < 17; ++k) V[k] = sC[q][y0 + k][x];
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    float a = 0.f;
#pragma unroll
                    for (int k = 0; k < 11; ++k) a = fmaf(w[k], V[j + k], a);
                    o[j][q] = a;
                }
            }
            const int gx = tx0 + x - HALO;
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const int gy = ty0 + y0 + j - HALO;
                float val, d0, d1, d2;
                ssim_point4(o[j], C1, C2, val, d0, d1, d2);
                const bool in = gx >= 0 && gx < W && gy >= 0 && gy < H;
                const bool valid = in && gx >= crop && gx < W - crop && gy >= crop && gy < H - crop;
                const float ch = valid ? chain : 0.f;
                sD[0][y0 + j][x] = in ? d0 * ch : 0.f; sD[1][y0 + j][x] = in ? d1 * ch : 0.f; sD[2][y0 + j][x] = in ? d2 * ch : 0.f;
                const bool own = x >= HALO && x < HALO + T && y0 + j >= HALO && y0 + j < HALO + T;   // the tile's own pixels: the loss sum
                ss += (valid && own) ? val : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) l1 += (tx0 + ox < W && ty0 + oy0 + j < H) ? fabsf(p1[j] - p2[j]) : 0.f;
        __syncthreads();
        // ---- second horizontal pass over the maps: 42 rows x 32 columns, strips of 2 (16 per row: 672 items)
        for (int it = tid; it < R1 * (T / 2); it += NTS) {
            const int r = it % R1, x0 = (it / R1) * 2;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                float V[12];
#pragma unroll
                for (int k = 0; k < 12; ++k) V[k] = sD[q][r][x0 + k];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float a = 0.f;
#pragma unroll
                    for (int k = 0; k < 11; ++k) a = fmaf(w[k], V[j + k], a);
                    sE[q][r][x0 + j] = a;
                }
            }
        }
        __syncthreads();
        // ---- second vertical pass + the gradient of the tile's pixels
        {
            float sv[4][3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                float V[14];
#pragma unroll
                for (int k = 0; k < 14; ++k) V[k] = sE[q][oy0 + k][ox];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float a = 0.f;
#pragma unroll
                    for (int k = 0; k < 11; ++k) a = fmaf(w[k], V[j + k], a);
                    sv[j][q] = a;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gx = tx0 + ox, gy = ty0 + oy0 + j;
                float v = 0.f;
                if (gx < W && gy < H) {
                    const float raw = rbase[((size_t)gy * W + gx) * 3];
                    const float d = p1[j] - p2[j];
                    const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
                    v = sv[j][0] + (2.f * p1[j]) * sv[j][1] + p2[j] * sv[j][2] + l1_coeff * sgn;
                    v = (raw >= 0.f && raw <= 1.f) ? v * up : 0.f;
                }
                g[j][c] = v;
            }
        }
        __syncthreads();   // the next channel restages over the maps / planes
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int gx = tx0 + ox, gy = ty0 + oy0 + j;
        if (gx < W && gy < H) {
            float* o = v_render + ((size_t)blockIdx.z * npix + (size_t)gy * W + gx) * 3;
            o[0] = g[j][0]; o[1] = g[j][1]; o[2] = g[j][2];
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { l1 += __shfl_xor(l1, m); ss += __shfl_xor(ss, m); }
    if (lane == 0) s_red[wave] = make_float2(l1, ss);
    __syncthreads();
    if (tid == 0) {
        float2 t = s_red[0];
#pragma unroll
        for (int k = 1; k < NTS / 64; ++k) { t.x += s_red[k].x; t.y += s_red[k].y; }
        block_sums[(size_t)blockIdx.z * n + tix] = t;
    }
}

inline dim3 tile_grid(uint32_t B, uint32_t H, uint32_t W) { return dim3((W + TX - 1) / TX, (H + TY - 1) / TY, B); }
// fused loss kernels: 1-D XCD-aware grid (block b runs on XCD b % 8): every XCD takes a contiguous row-major run of tiles, so that the halo
// lines neighbouring tiles share are fetched into ONE L2 instead of two (a 74-float row of a 64-pixel tile touches four 128 B lines, two of
// them shared with its neighbours: with tiles dealt round-robin to the XCDs the loss kernels fetched 2.0x / 2.1x their algorithmic bytes).
// Same-box A/B at 1080p (tools/loss_ab.py, each kernel on its own): forward 0.0607 -> 0.0558 ms, backward 0.0512 -> 0.0444; inside the
// training step (bench.py per-op events) 0.064 + 0.054 -> 0.061 + 0.047.  (Round 2 measured an XCD-aware order as slower; that was with
// the 64 x 16 tiles.)
inline dim3 loss_grid(uint32_t B, uint32_t H, uint32_t W) { const uint32_t n = ((W + TX - 1) / TX) * ((H + TY - 1) / TY); return dim3(((n + 7u) / 8u) * 8u, 1, B); }

}  // namespace ssim
}  // namespace gsx

using namespace gsx;
using namespace gsx::ssim;

extern "C" int gsx_fused_ssim_fwd(uint32_t B, uint32_t CH, uint32_t H, uint32_t W, float C1, float C2, const float* img1, const float* img2,
                                  float* ssim_map, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12, void* stream) {
    if (B == 0 || CH == 0 || H == 0 || W == 0) return GSX_OK;
    if (!img1 || !img2 || !ssim_map) { set_error("gsx_fused_ssim_fwd: null pointer"); return GSX_ERR_INVALID_ARGUMENT; }
    const bool train = dm_dmu1 != nullptr;
    if (train && (!dm_dsigma1_sq || !dm_dsigma12)) { set_error("gsx_fused_ssim_fwd: all three derivative maps or none"); return GSX_ERR_INVALID_ARGUMENT; }
    if (B > 65535u) { set_error("gsx_fused_ssim_fwd: batch above 65535"); return GSX_ERR_UNSUPPORTED; }
    hipStream_t st = (hipStream_t)stream;
    if (train)
        hipLaunchKernelGGL(ssim_fwd_kernel<true>, tile_grid(B, H, W), dim3(NT), 0, st, (int)CH, (int)H, (int)W, C1, C2, img1, img2, ssim_map,
                           dm_dmu1, dm_dsigma1_sq, dm_dsigma12);
    else
        hipLaunchKernelGGL(ssim_fwd_kernel<false>, tile_grid(B, H, W), dim3(NT), 0, st, (int)CH, (int)H, (int)W, C1, C2, img1, img2, ssim_map,
                           nullptr, nullptr, nullptr);
    return check_launch("gsx_fused_ssim_fwd");
}

extern "C" int gsx_fused_ssim_bwd(uint32_t B, uint32_t CH, uint32_t H, uint32_t W, float C1, float C2, const float* img1, const float* img2,
                                  const float* dL_dmap, float* dL_dimg1, const float* dm_dmu1, const float* dm_dsigma1_sq,
                                  const float* dm_dsigma12, void* stream) {
    (void)C1;
    (void)C2;
    if (B == 0 || CH == 0 || H == 0 || W == 0) return GSX_OK;
    if (!img1 || !img2 || !dL_dmap || !dL_dimg1 || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12) {
        set_error("gsx_fused_ssim_bwd: null pointer");
        return GSX_ERR_INVALID_ARGUMENT;
    }
    if (B > 65535u) { set_error("gsx_fused_ssim_bwd: batch above 65535"); return GSX_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL(ssim_bwd_kernel, tile_grid(B, H, W), dim3(NT), 0, (hipStream_t)stream, (int)CH, (int)H, (int)W, img1, img2, dL_dmap,
                       dL_dimg1, dm_dmu1, dm_dsigma1_sq, dm_dsigma12);
    return check_launch("gsx_fused_ssim_bwd");
}

static size_t loss_maps_bytes(uint32_t C, uint32_t H, uint32_t W) { return (size_t)9 * C * H * W * sizeof(float); }
static uint32_t loss_blocks(uint32_t C, uint32_t H, uint32_t W) { return C * loss_blocks_per_image((int)H, (int)W); }

extern "C" size_t gsx_photometric_loss_workspace_bytes(uint32_t C, uint32_t H, uint32_t W) {
    return loss_maps_bytes(C, H, W) + (size_t)loss_blocks(C, H, W) * sizeof(float2);
}

extern "C" int gsx_photometric_loss_fwd(uint32_t C, uint32_t H, uint32_t W, float lambda_dssim, const float* render, const float* gt,
                                        float* loss3, void* workspace, size_t workspace_bytes, void* stream) {
    if (C == 0 || H == 0 || W == 0) { set_error("gsx_photometric_loss_fwd: empty image"); return GSX_ERR_INVALID_ARGUMENT; }
    if (!render || !gt || !loss3 || !workspace) { set_error("gsx_photometric_loss_fwd: null pointer"); return GSX_ERR_INVALID_ARGUMENT; }
    if (workspace_bytes < gsx_photometric_loss_workspace_bytes(C, H, W)) { set_error("gsx_photometric_loss_fwd: workspace too small"); return GSX_ERR_WORKSPACE_TOO_SMALL; }
    if (C > 65535u) { set_error("gsx_photometric_loss_fwd: more than 65535 images"); return GSX_ERR_UNSUPPORTED; }
    // "valid" padding crops 5 px of the map unless the image is too small (fused_ssim.cuh:61-67)
    const int crop = (H > 10 && W > 10) ? 5 : 0;
    const double n_valid = (double)C * 3.0 * (double)(H - 2 * crop) * (double)(W - 2 * crop);
    const double n_all = (double)C * 3.0 * (double)H * (double)W;
    float* maps = (float*)workspace;
    float2* sums = (float2*)((char*)workspace + loss_maps_bytes(C, H, W));
    hipStream_t st = (hipStream_t)stream;
    // Upstream's backward scatters dL/dmap into a zero image only when it cropped (fused_ssim.cuh:85-96): for images of
    // 10 px or less the SSIM term contributes no gradient.  Kept.
    const float chain = crop ? (float)(-(double)lambda_dssim / n_valid) : 0.f;
    hipLaunchKernelGGL(loss_fwd_kernel, loss_grid(C, H, W), dim3(NT), 0, st, (int)H, (int)W, chain, crop, render, gt, maps, sums);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(NT), 0, st, loss_blocks(C, H, W), (const float2*)sums, 1.0 / n_all, 1.0 / n_valid,
                       lambda_dssim, loss3);
    return check_launch("gsx_photometric_loss_fwd");
}

// ABI 7: loss3 AND v_render = d loss / d render * grad_scale in ONE kernel (loss_single_pass_kernel); workspace: the block sums only.
extern "C" size_t gsx_photometric_loss_single_pass_workspace_bytes(uint32_t C, uint32_t H, uint32_t W) {
    return (size_t)C * ((W + sp::T - 1) / sp::T) * ((H + sp::T - 1) / sp::T) * sizeof(float2);
}
extern "C" int gsx_photometric_loss_single_pass(uint32_t C, uint32_t H, uint32_t W, float lambda_dssim, float grad_scale, const float* render, const float* gt,
                                                float* loss3, float* v_render, void* workspace, size_t workspace_bytes, void* stream) {
    if (C == 0 || H == 0 || W == 0) { set_error("gsx_photometric_loss_single_pass: empty image"); return GSX_ERR_INVALID_ARGUMENT; }
    if (!render || !gt || !loss3 || !v_render || !workspace) { set_error("gsx_photometric_loss_single_pass: null pointer"); return GSX_ERR_INVALID_ARGUMENT; }
    if (workspace_bytes < gsx_photometric_loss_single_pass_workspace_bytes(C, H, W)) { set_error("gsx_photometric_loss_single_pass: workspace too small"); return GSX_ERR_WORKSPACE_TOO_SMALL; }
    if (C > 65535u) { set_error("gsx_photometric_loss_single_pass: more than 65535 images"); return GSX_ERR_UNSUPPORTED; }
    const int crop = (H > 10 && W > 10) ? 5 : 0;
    const double n_valid = (double)C * 3.0 * (double)(H - 2 * crop) * (double)(W - 2 * crop), n_all = (double)C * 3.0 * (double)H * (double)W;
    const float chain = crop ? (float)(-(double)lambda_dssim / n_valid) : 0.f;   // (fused_ssim.cuh:85-96: no SSIM gradient for images of 10 px or less)
    const uint32_t n_tiles = ((W + sp::T - 1) / sp::T) * ((H + sp::T - 1) / sp::T);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(loss_single_pass_kernel, dim3(((n_tiles + 7u) / 8u) * 8u, 1, C), dim3(sp::NTS), 0, st, (int)H, (int)W, chain, crop,
                       (float)((1.0 - (double)lambda_dssim) / n_all), grad_scale, render, gt, v_render, (float2*)workspace);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(NT), 0, st, C * n_tiles, (const float2*)workspace, 1.0 / n_all, 1.0 / n_valid, lambda_dssim, loss3);
    return check_launch("gsx_photometric_loss_single_pass");
}

extern "C" int gsx_photometric_loss_bwd(uint32_t C, uint32_t H, uint32_t W, float lambda_dssim, const float* grad_loss, float grad_scale,
                                        const float* render, const float* gt, const void* workspace, size_t workspace_bytes,
                                        float* v_render, void* stream) {
    if (C == 0 || H == 0 || W == 0) { set_error("gsx_photometric_loss_bwd: empty image"); return GSX_ERR_INVALID_ARGUMENT; }
    if (!render || !gt || !workspace || !v_render) { set_error("gsx_photometric_loss_bwd: null pointer"); return GSX_ERR_INVALID_ARGUMENT; }
    if (workspace_bytes < gsx_photometric_loss_workspace_bytes(C, H, W)) { set_error("gsx_photometric_loss_bwd: workspace too small"); return GSX_ERR_WORKSPACE_TOO_SMALL; }
    if (C > 65535u) { set_error("gsx_photometric_loss_bwd: more than 65535 images"); return GSX_ERR_UNSUPPORTED; }
    const double n_all = (double)C * 3.0 * (double)H * (double)W;
    hipLaunchKernelGGL(loss_bwd_kernel, loss_grid(C, H, W), dim3(NT), 0, (hipStream_t)stream, (int)H, (int)W,
                       (float)((1.0 - (double)lambda_dssim) / n_all), grad_loss, grad_scale, render, gt, (const float*)workspace, v_render);
    return check_launch("gsx_photometric_loss_bwd");
}
