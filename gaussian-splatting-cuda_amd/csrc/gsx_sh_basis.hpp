// gsx_sh_basis.hpp — real spherical-harmonics basis up to degree 4 at a unit direction (Sloan, "Efficient Spherical Harmonic
// Evaluation", JCGT 2013; reference: gsplat/SphericalHarmonicsCUDA.cu:20-371), shared by gsx_sh.hip and the fused front end.
#pragma once
#include "gsx_device.hpp"

namespace gsx {

template <int DEG> struct ShBasis {
    // Y[k] for k < (DEG+1)^2 at unit direction (x,y,z); optionally the partials wrt x,y,z.
    template <bool GRAD>
    GSX_DEV static void eval(float x, float y, float z, float* Y, float* Yx, float* Yy, float* Yz) {
        constexpr int NB = (DEG + 1) * (DEG + 1);
        if (GRAD) {
#pragma unroll
            for (int k = 0; k < NB; ++k) { Yx[k] = 0.f; Yy[k] = 0.f; Yz[k] = 0.f; }
        }
        Y[0] = 0.2820947917738781f;
        if (DEG >= 1) {
            const float c1 = 0.48860251190292f;
            Y[1] = -c1 * y; Y[2] = c1 * z; Y[3] = -c1 * x;
            if (GRAD) { Yy[1] = -c1; Yz[2] = c1; Yx[3] = -c1; }
        }
        float z2 = 0.f, fC1 = 0.f, fS1 = 0.f, fC1_x = 0.f, fC1_y = 0.f, fS1_x = 0.f, fS1_y = 0.f;
        if (DEG >= 2) {
            z2 = z * z;
            const float a = -1.092548430592079f;
            const float fT0B = a * z;
            fC1 = x * x - y * y; fS1 = 2.f * x * y;
            const float c2 = 0.5462742152960395f;
            Y[4] = c2 * fS1; Y[5] = fT0B * y; Y[6] = 0.9461746957575601f * z2 - 0.3153915652525201f;
            Y[7] = fT0B * x; Y[8] = c2 * fC1;
            if (GRAD) {
                fC1_x = 2.f * x; fC1_y = -2.f * y; fS1_x = 2.f * y; fS1_y = 2.f * x;
                Yx[4] = c2 * fS1_x; Yy[4] = c2 * fS1_y;
                Yy[5] = fT0B; Yz[5] = a * y;
                Yz[6] = 2.f * 0.9461746957575601f * z;
                Yx[7] = fT0B; Yz[7] = a * x;
                Yx[8] = c2 * fC1_x; Yy[8] = c2 * fC1_y;
            }
        }
        float fC2 = 0.f, fS2 = 0.f, fC2_x = 0.f, fC2_y = 0.f, fS2_x = 0.f, fS2_y = 0.f, Y12_z = 0.f;
        if (DEG >= 3) {
            const float fT0C = -2.285228997322329f * z2 + 0.4570457994644658f;
            const float b = 1.445305721320277f;
            const float fT1B = b * z;
            fC2 = x * fC1 - y * fS1; fS2 = x * fS1 + y * fC1;
            const float c3 = -0.5900435899266435f;
            Y[9] = c3 * fS2; Y[10] = fT1B * fS1; Y[11] = fT0C * y;
            Y[12] = z * (1.865881662950577f * z2 - 1.119528997770346f);
            Y[13] = fT0C * x; Y[14] = fT1B * fC1; Y[15] = c3 * fC2;
            if (GRAD) {
                const float fT0C_z = -2.285228997322329f * 2.f * z;
                fC2_x = fC1 + x * fC1_x - y * fS1_x; fC2_y = x * fC1_y - fS1 - y * fS1_y;
                fS2_x = fS1 + x * fS1_x + y * fC1_x; fS2_y = x * fS1_y + fC1 + y * fC1_y;
                Y12_z = 3.f * 1.865881662950577f * z2 - 1.119528997770346f;
                Yx[9] = c3 * fS2_x; Yy[9] = c3 * fS2_y;
                Yx[10] = fT1B * fS1_x; Yy[10] = fT1B * fS1_y; Yz[10] = b * fS1;
                Yy[11] = fT0C; Yz[11] = fT0C_z * y;
                Yz[12] = Y12_z;
                Yx[13] = fT0C; Yz[13] = fT0C_z * x;
                Yx[14] = fT1B * fC1_x; Yy[14] = fT1B * fC1_y; Yz[14] = b * fC1;
                Yx[15] = c3 * fC2_x; Yy[15] = c3 * fC2_y;
            }
        }
        if (DEG >= 4) {
            const float fT0D = z * (-4.683325804901025f * z2 + 2.007139630671868f);
            const float fT1C = 3.31161143515146f * z2 - 0.47308734787878f;
            const float d = -1.770130769779931f;
            const float fT2B = d * z;
            const float fC3 = x * fC2 - y * fS2, fS3 = x * fS2 + y * fC2;
            const float c4 = 0.6258357354491763f;
            Y[16] = c4 * fS3; Y[17] = fT2B * fS2; Y[18] = fT1C * fS1; Y[19] = fT0D * y;
            Y[20] = 1.984313483298443f * z * Y[12] - 1.006230589874905f * Y[6];
            Y[21] = fT0D * x; Y[22] = fT1C * fC1; Y[23] = fT2B * fC2; Y[24] = c4 * fC3;
            if (GRAD) {
                const float fT0D_z = 3.f * -4.683325804901025f * z2 + 2.007139630671868f;
                const float fT1C_z = 2.f * 3.31161143515146f * z;
                const float fC3_x = fC2 + x * fC2_x - y * fS2_x, fC3_y = x * fC2_y - fS2 - y * fS2_y;
                const float fS3_x = fS2 + y * fC2_x + x * fS2_x, fS3_y = x * fS2_y + fC2 + y * fC2_y;
                Yx[16] = c4 * fS3_x; Yy[16] = c4 * fS3_y;
                Yx[17] = fT2B * fS2_x; Yy[17] = fT2B * fS2_y; Yz[17] = d * fS2;
                Yx[18] = fT1C * fS1_x; Yy[18] = fT1C * fS1_y; Yz[18] = fT1C_z * fS1;
                Yy[19] = fT0D; Yz[19] = fT0D_z * y;
                Yz[20] = 1.984313483298443f * (Y[12] + z * Y12_z) - 1.006230589874905f * Yz[6];
                Yx[21] = fT0D; Yz[21] = fT0D_z * x;
                Yx[22] = fT1C * fC1_x; Yy[22] = fT1C * fC1_y; Yz[22] = fT1C_z * fC1;
                Yx[23] = fT2B * fC2_x; Yy[23] = fT2B * fC2_y; Yz[23] = d * fC2;
                Yx[24] = c4 * fC3_x; Yy[24] = c4 * fC3_y;
            }
        }
    }
};

// camera position -R^T t of a row-major world->camera [4,4] (exact for a rigid transform; upstream uses torch::inverse)
GSX_DEV f3 cam_position(const float* __restrict__ vm) {
    const f3 t{vm[3], vm[7], vm[11]};
    return {-(vm[0] * t.x + vm[4] * t.y + vm[8] * t.z), -(vm[1] * t.x + vm[5] * t.y + vm[9] * t.z),
            -(vm[2] * t.x + vm[6] * t.y + vm[10] * t.z)};
}

}  // namespace gsx
