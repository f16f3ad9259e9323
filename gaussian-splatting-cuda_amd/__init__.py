"""gsx — MI355X-native differentiable 3DGS rasterizer hot path (the `--gut` / gsplat::Ops.h path of
MrNeRF/gaussian-splatting-cuda), hand-written HIP for gfx950 behind a C ABI.

    ops         the seven gsplat:: operators (C++ shim over libgsx.so, same names/signatures as gsplat/Ops.h)
    rasterizer  gs::training::rasterize glue + autograd wrappers (rasterizer.cpp / rasterizer_autograd.cpp)
    scenes      synthetic scenes of BASELINE.json's configs
    distributed camera-sharded data parallelism (one camera per rank, RCCL gradient all-reduce)

The directory name contains a hyphen; import it with importlib.import_module("gaussian-splatting-cuda_amd")
or through the top-level alias module `gsx`.
"""
__all__ = ["ops", "rasterizer", "scenes", "distributed", "build"]
