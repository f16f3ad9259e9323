"""Debug tool (round 5): the shader clock the blend kernels really run at, read INSIDE the kernels.  Builds libgsx with -DGSX_CLOCKS into
gpurun_out/ (thread 0 of every block brackets its block with s_memtime = shader cycles and s_memrealtime = 100 MHz; nothing else is
counted, so the kernels run at production speed), runs the blend forward + backward of S-1M (or S-5M) through tools/blend_ab.py and prints,
per kernel, sum(cycles) / sum(10 ns ticks) = the residency-weighted clock.  With `stats` as second argument: a -DGSX_STATS build instead
(work counters with global atomics: the kernels run 10 - 400x slower) for the Gaussian-major backward's flush-lock spins per pass.
Run on the GPU box: python tools/blend_clock.py [1m|5m] [stats]"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401


def main():
    csrc = os.path.join(ROOT, "gaussian-splatting-cuda_amd", "csrc")
    stats = len(sys.argv) > 2 and sys.argv[2] == "stats"
    out = os.path.join(ROOT, "gpurun_out", "libgsx_stats.so" if stats else "libgsx_clocks.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    from importlib import import_module
    srcs = [os.path.join(csrc, f) for f in import_module("gaussian-splatting-cuda_amd.build").HIP_SOURCES]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize",
                           "-DGSX_STATS" if stats else "-DGSX_CLOCKS", "-o", out] + srcs)
    lib = ctypes.CDLL(out, mode=ctypes.RTLD_GLOBAL)   # preloaded under the same soname: the extension binds to this build
    which = sys.argv[1] if len(sys.argv) > 1 else "1m"
    sys.argv = [sys.argv[0], which, "10"]
    clk = (ctypes.c_ulonglong * 12)()
    st = (ctypes.c_ulonglong * 16)()
    import runpy
    if stats:
        lib.gsx_debug_read_stats(st, 1)
    else:
        lib.gsx_debug_read_clocks(clk, 1)
    runpy.run_path(os.path.join(ROOT, "tools", "blend_ab.py"), run_name="__main__")
    if stats:
        lib.gsx_debug_read_stats(st, 1)
    else:
        lib.gsx_debug_read_clocks(clk, 1)
    for k, name in enumerate(["raster_fwd_fast_kernel (one list)", "raster_fwd_quad_kernel (four lists)", "raster_bwd_gq_kernel (Gaussian-major)", "raster_bwd_fast_kernel (pixel-major)"]):
        cyc, ticks, blocks = clk[3 * k], clk[3 * k + 1], clk[3 * k + 2]
        if blocks:
            print("%-40s blocks %8d  mean block residency %8.1f us  shader clock %.3f GHz" % (name, blocks, ticks / blocks / 100.0, cyc / (ticks * 10.0)))
    if st[9]:
        print("raster_bwd_gq_kernel: flush-lock spins per pass %.3f (%d spins / %d passes)" % (st[12] / st[9], st[12], st[9]))


if __name__ == "__main__":
    main()
