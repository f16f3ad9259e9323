"""Debug tool: build libgsx with -DGSX_STATS into gpurun_out/, run the S-1M forward once and print the
culling / early-exit counters of the fast blend kernel.  Run on the GPU box: python tools/fwd_stats.py [1m|5m]"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    csrc = os.path.join(ROOT, "gaussian-splatting-cuda_amd", "csrc")
    out = os.path.join(ROOT, "gpurun_out", "libgsx_stats.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    srcs = [os.path.join(csrc, f) for f in ["gsx_capi.hip", "gsx_sh.hip", "gsx_projection.hip", "gsx_intersect.hip",
                                             "gsx_raster.hip", "gsx_raster_fast.hip", "gsx_frontend.hip", "gsx_mcmc.hip", "gsx_adam.hip", "gsx_ssim.hip"]]
    prebuilt = os.path.join(ROOT, "tools", "variants", "libgsx_stats.so")   # bash tools/build_variant.sh stats -DGSX_STATS (CPU side: saves two minutes of box time)
    if os.path.exists(prebuilt):
        out = prebuilt
    else:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                               "-fno-slp-vectorize", "-DGSX_STATS", "-o", out] + srcs)
    # swap the library the extension binds to: preload the stats build under the same soname
    lib = ctypes.CDLL(out, mode=ctypes.RTLD_GLOBAL)
    import gsx  # noqa: F401
    from gsx import rasterizer, scenes
    dev = "cuda:0"
    scene = {"1m": scenes.scene_1m, "5m": scenes.scene_5m}[sys.argv[1] if len(sys.argv) > 1 else "1m"]()
    model = scenes.to_splat_data(scene, dev)
    cam = rasterizer.Camera(viewmat=scene["viewmat"].to(dev), K=scene["K"].to(dev), width=scene["width"], height=scene["height"])
    with torch.no_grad():
        o = rasterizer.rasterize(cam, model, scene["background"].to(dev))
    buf = (ctypes.c_ulonglong * 16)()   # gsx_debug_read_stats copies all 16 counters
    lib.gsx_debug_read_stats(buf, 1)
    names = ["wave steps (one-list kernel: wave-Gaussian evaluations; four-list kernel: compositing steps)", "cull candidates (wave x Gaussian)",
             "steps with >=1 contributing lane", "contributing (pixel,Gaussian) pairs",
             "four-list kernel: sum of the four lists' lengths", "four-list kernel: steps if the lists ran on across chunk boundaries",
             "pair kernel: stop branches taken (wave x step)", "pair kernel: pixels stopped in them"]
    I = o.n_isects
    print("n_isects", I, " 4*I =", 4 * I)
    for n, v in zip(names, buf):
        if n:
            print("%-45s %12d  (%.3f per isect)" % (n, v, v / I))


if __name__ == "__main__":
    main()
