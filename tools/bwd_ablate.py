"""Debug tool (GPU box): time the S-1M backward blend kernel for ablated builds (-DGSX_ABLATE=n) of libgsx.
  0 = product kernel, 1 = no cross-lane reduction, 2 = butterfly but no LDS atomics, 3 = no finishing step, 4 = finishing math without the global atomics.
Each variant runs in its own process (the preloaded library shadows libgsx.so's symbols)."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(variant):
    import torch
    csrc = os.path.join(ROOT, "gaussian-splatting-cuda_amd", "csrc")
    out = os.path.join(ROOT, "gpurun_out", "libgsx_abl%d.so" % variant)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    srcs = [os.path.join(csrc, f) for f in ["gsx_capi.hip", "gsx_sh.hip", "gsx_projection.hip", "gsx_intersect.hip",
                                             "gsx_raster.hip", "gsx_raster_fast.hip"]]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize",
                           "-DGSX_ABLATE=%d" % variant, "-o", out] + srcs)
    ctypes.CDLL(out, mode=ctypes.RTLD_GLOBAL)
    import gsx  # noqa: F401
    from gsx import rasterizer, scenes
    dev = "cuda:0"
    scene = scenes.scene_1m()
    model = scenes.to_splat_data(scene, dev)
    for p in model.params():
        p.requires_grad_(True)
    cam = rasterizer.Camera(viewmat=scene["viewmat"].to(dev), K=scene["K"].to(dev), width=scene["width"], height=scene["height"])
    bg = scene["background"].to(dev)
    from gsx import ops
    times = []
    orig = ops.rasterize_to_pixels_from_world_3dgs_bwd

    def timed(*a):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); r = orig(*a); e.record()
        times.append((s, e))
        return r
    ops.rasterize_to_pixels_from_world_3dgs_bwd = timed
    for _ in range(6):
        out_ = rasterizer.rasterize(cam, model, bg)
        out_.image.sum().backward()
    torch.cuda.synchronize()
    ms = [s.elapsed_time(e) for s, e in times[2:]]
    print("GSX_ABLATE=%d  backward blend %.4f ms" % (variant, sum(ms) / len(ms)))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(int(sys.argv[1]))
    else:
        for v in (0, 3, 4):
            subprocess.call([sys.executable, os.path.abspath(__file__), str(v)])
