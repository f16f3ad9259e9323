// compat/gsplat/Projection.h — stands in for the reference's gsplat/Projection.h.  The only caller-side include
// (src/training/rasterization/rasterizer_autograd.cpp:6) wants the types of Cameras.h; the kernel launcher the reference header
// declares (`launch_projection_ut_3dgs_fused_kernel`, Projection.h:12-41) is internal to its CUDA backend and has no counterpart
// here: `gsplat::projection_ut_3dgs_fused` (Ops.h) goes straight to the C ABI (`gsx_projection_ut_3dgs_fused`, include/gsx.h).
#pragma once

#include "Cameras.h"
#include <cstdint>
