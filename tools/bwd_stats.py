"""Debug tool: build libgsx with -DGSX_STATS into gpurun_out/, run the blend forward + backward of S-1M (or S-5M) once and print the
work counters of the Gaussian-major backward kernel (raster_bwd_gq_kernel).  Run on the GPU box: python tools/bwd_stats.py [1m|5m]"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401


def main():
    csrc = os.path.join(ROOT, "gaussian-splatting-cuda_amd", "csrc")
    out = os.path.join(ROOT, "gpurun_out", "libgsx_stats.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    srcs = [os.path.join(csrc, f) for f in ["gsx_capi.hip", "gsx_sh.hip", "gsx_projection.hip", "gsx_intersect.hip",
                                             "gsx_raster.hip", "gsx_raster_fast.hip", "gsx_mcmc.hip", "gsx_adam.hip", "gsx_ssim.hip"]]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize",
                           "-DGSX_STATS", "-o", out] + srcs)
    lib = ctypes.CDLL(out, mode=ctypes.RTLD_GLOBAL)   # preloaded under the same soname: the extension binds to this build
    which = sys.argv[1] if len(sys.argv) > 1 else "1m"
    sys.argv = [sys.argv[0], which, "1"]
    buf = (ctypes.c_ulonglong * 16)()
    import runpy
    lib.gsx_debug_read_stats(buf, 1)
    runpy.run_path(os.path.join(ROOT, "tools", "blend_ab.py"), run_name="__main__")   # 3 warm-up + 1 timed launch of each op = 4 launches
    lib.gsx_debug_read_stats(buf, 1)
    n = 4.0
    print("per launch: list entries (4x4 block, Gaussian) %.3fM  passes of 16 Gaussians x 16 pixels %.1fK (= %.3fM lane-pixel slots)  super-chunks %.1fK  staged %.3fM" %
          (buf[8] / n / 1e6, buf[9] / n / 1e3, buf[9] / n * 256 / 1e6, buf[10] / n / 1e3, buf[11] / n / 1e6))
    print("pass fill %.3f" % (buf[8] / max(1, buf[9] * 16)))


if __name__ == "__main__":
    main()
