// compat/gsplat/Common.h — stands in for the reference's gsplat/Common.h (lines 1-55) when its `src/` is built against the
// MI355X backend.  Put `compat/gsplat` on the include path INSTEAD of the reference's `gsplat/` directory: the reference TUs
// that say `#include "Common.h"` (include/core/camera.hpp:7, src/loader/formats/colmap.hpp:6) then get exactly one definition
// of `gsplat::CameraModelType`.
//
// Differences from the reference header, all invisible to its callers under src/ and include/:
//   * no glm: the `vec2 … mat3x2` typedefs (Common.h:35-41) are only used by the reference's own .cu files, which this
//     backend replaces; no caller TU names `gsplat::vec*` / `gsplat::mat*`;
//   * CUB_WRAPPER (Common.h:23-30) is dropped for the same reason; DEVICE_GUARD is spelled with the HIP guard.
#pragma once

#include <algorithm>
#include <cstdint>

namespace gsplat {

#define CHECK_CUDA(x) TORCH_CHECK(x.is_cuda(), #x " must be a CUDA tensor")
#define CHECK_CONTIGUOUS(x) TORCH_CHECK(x.is_contiguous(), #x " must be contiguous")
#define CHECK_INPUT(x) \
    CHECK_CUDA(x);     \
    CHECK_CONTIGUOUS(x)
#define DEVICE_GUARD(_ten) const c10::hip::OptionalHIPGuardMasqueradingAsCUDA device_guard(device_of(_ten));

    // gsplat/Common.h:46-50
    enum CameraModelType {
        PINHOLE = 0,
        ORTHO = 1,
        FISHEYE = 2,
    };

#define N_THREADS_PACKED 256
#define ALPHA_THRESHOLD  (1.f / 255.f)

} // namespace gsplat
