// compat/gsplat/Cameras.h — stands in for the host-visible part of the reference's gsplat/Cameras.h (lines 1-61): the shutter
// enum and the unscented-transform parameter block with its tensor (de)serialisation, which the autograd wrappers use to park the
// parameters in `ctx->saved_data` (src/training/rasterization/rasterizer_autograd.cpp:290 `ut_params.to_tensor()`, :364
// `UnscentedTransformParameters::from_tensor(...)`).  The rest of the reference header (Cameras.h:62-…, glm based camera parameter
// structs) is device-side material of the CUDA kernels this backend replaces; no TU under src/ or include/ names it.
#pragma once

#include <ATen/ATen.h>
#include <c10/util/Exception.h>

#include <cstdint>

// gsplat/Cameras.h:16-22 (order matters: the integer values cross the C ABI as gsx_shutter)
enum class ShutterType {
    ROLLING_TOP_TO_BOTTOM,
    ROLLING_LEFT_TO_RIGHT,
    ROLLING_BOTTOM_TO_TOP,
    ROLLING_RIGHT_TO_LEFT,
    GLOBAL
};

// gsplat/Cameras.h:27-61
struct UnscentedTransformParameters {
    float alpha = 0.1f;                          // sigma-point spread (Wan & van der Merwe 2000)
    float beta = 2.f;
    float kappa = 0.f;
    float in_image_margin_factor = 0.1f;         // sigma points may land this fraction of the image size outside it
    bool require_all_sigma_points_valid = true;  // false: one valid sigma point is enough

    // 1-D float32 CPU tensor [alpha, beta, kappa, margin, require_all] (Cameras.h:46-50)
    at::Tensor to_tensor() const {
        const float v[5] = {alpha, beta, kappa, in_image_margin_factor, require_all_sigma_points_valid ? 1.f : 0.f};
        return at::tensor(at::ArrayRef<float>(v, 5), at::TensorOptions().dtype(at::kFloat));
    }

    // inverse of to_tensor (Cameras.h:52-60); same error text
    static UnscentedTransformParameters from_tensor(const at::Tensor& tensor) {
        TORCH_CHECK(tensor.dim() == 1 && tensor.size(0) == 5, "UnscentedTransformParameters must be a 1D tensor of size 5");
        UnscentedTransformParameters p;
        p.alpha = tensor[0].item<float>();
        p.beta = tensor[1].item<float>();
        p.kappa = tensor[2].item<float>();
        p.in_image_margin_factor = tensor[3].item<float>();
        p.require_all_sigma_points_valid = tensor[4].item<bool>();
        return p;
    }
};
