"""In-tree build of the native pieces (no JIT cache: the .so files must travel with the repo snapshot).

  libgsx.so                      hipcc --offload-arch=gfx950  csrc/*.hip      (kernels + C ABI, no torch)
  _gsx_ops.<abi>.so              g++                           csrc/ops_shim.cpp (namespace gsplat on at::Tensor + pybind11)
  libgsx_gsplat_backend.so       g++ -DGSX_NO_PYBIND           csrc/ops_shim.cpp (the same shim without Python: what a reference build links
                                                                instead of its `gsplat_backend` static library)

hipcc is invoked directly (not torch.utils.cpp_extension, which would run hipify over the sources).
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIBGSX = os.path.join(HERE, "libgsx.so")
HIP_SOURCES = ["gsx_capi.hip", "gsx_sh.hip", "gsx_projection.hip", "gsx_intersect.hip", "gsx_raster.hip", "gsx_raster_fast.hip", "gsx_frontend.hip", "gsx_mcmc.hip", "gsx_adam.hip", "gsx_ssim.hip"]
HIP_HEADERS = ["gsx_device.hpp", "gsx_raster_common.hpp", "gsx_record.hpp", "gsx_sh_basis.hpp", "gsx_ut_project.hpp"]


def blend_kernel_hash():
    """sha256 over the sources the blend kernels are compiled from (+ the compile flags): the committed counter-derived figures of
    profiles/pmc.json carry the hash they were measured with, and bench.py reports them only while it still matches."""
    import hashlib
    h = hashlib.sha256()
    for f in ("gsx_raster_fast.hip", "gsx_raster_common.hpp", "gsx_record.hpp", "gsx_device.hpp"):
        h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(("-O3 -fno-slp-vectorize " + os.environ.get("GSX_EXTRA_HIPCC_FLAGS", "")).encode())
    return h.hexdigest()[:16]


def ops_module_path():
    return os.path.join(HERE, "_gsx_ops" + sysconfig.get_config_var("EXT_SUFFIX"))


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        raise RuntimeError("build failed: " + " ".join(cmd[:4]) + " ...")
    return r.stdout


def build_libgsx(force=False, verbose=False):
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in HIP_HEADERS] + [os.path.join(INCLUDE, "gsx.h")]
    if not (force or _newer(LIBGSX, deps)):
        return LIBGSX
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for s in srcs:  # compile the translation units in parallel
        o = os.path.join(CSRC, os.path.basename(s) + ".o")
        objs.append(o)
        # -fno-slp-vectorize: v_pk_fma_f32 / v_pk_mul_f32 issue at ~7.5 cycles per wave64 instruction on gfx950 vs ~2.9 for the
        # scalar forms (tools/valu_probe.hip), so SLP-packed fp32 math is a net loss in the VALU-bound blend loops
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize"] + \
            os.environ.get("GSX_EXTRA_HIPCC_FLAGS", "").split() + ["-c", s, "-o", o]  # (tuning experiments: -DGSX_FCH=..., ...)
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out)
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
        if verbose and out.strip():
            print(out)
    _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIBGSX] + objs)
    return LIBGSX


COMPAT = os.path.join(os.path.dirname(HERE), "compat", "gsplat")
BACKEND = os.path.join(HERE, "libgsx_gsplat_backend.so")


def _shim_deps():
    src = os.path.join(CSRC, "ops_shim.cpp")
    return src, [src, os.path.join(INCLUDE, "gsx.h"), os.path.join(INCLUDE, "gsx_ops.h"), os.path.join(INCLUDE, "gsx_training_ops.h")] + \
        [os.path.join(COMPAT, h) for h in ("Ops.h", "Cameras.h", "Common.h", "Projection.h")]


def torch_cxx_flags():
    """Include / link flags of a TU that uses at::Tensor against this box's torch wheel (shared by the shim builds and the
    reference-call-site compile test)."""
    import torch
    tp = os.path.dirname(torch.__file__)
    inc = ["-std=c++17", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-D_GLIBCXX_USE_CXX11_ABI=1",
           "-I" + os.path.join(tp, "include"), "-I" + os.path.join(tp, "include", "torch", "csrc", "api", "include"), "-I/opt/rocm/include"]
    link = ["-L" + os.path.join(tp, "lib"), "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_hip", "-Wl,-rpath," + os.path.join(tp, "lib")]
    return inc, link


def build_backend_lib(force=False):
    """ops_shim.cpp without the pybind11 module: `namespace gsplat` (+ the Adam / SSIM drop-ins) as a plain shared library."""
    src, deps = _shim_deps()
    if not (force or _newer(BACKEND, deps)):
        return BACKEND
    inc, link = torch_cxx_flags()
    _run(["g++", "-O2", "-fPIC", "-shared", "-DGSX_NO_PYBIND=1"] + inc + [src, "-o", BACKEND, "-L" + HERE, "-lgsx"] + link + ["-Wl,-rpath,$ORIGIN"])
    return BACKEND


def build_ops_module(force=False):
    import pybind11
    import torch
    out = ops_module_path()
    src, deps = _shim_deps()
    if not (force or _newer(out, deps)):
        return out
    tp = os.path.dirname(torch.__file__)
    cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           "-DTORCH_EXTENSION_NAME=_gsx_ops", "-D_GLIBCXX_USE_CXX11_ABI=1",
           "-I" + os.path.join(tp, "include"), "-I" + os.path.join(tp, "include", "torch", "csrc", "api", "include"),
           "-I/opt/rocm/include", "-I" + sysconfig.get_paths()["include"], "-I" + pybind11.get_include(),
           src, "-o", out, "-L" + HERE, "-lgsx", "-L" + os.path.join(tp, "lib"),
           "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_hip", "-ltorch_python",
           "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(tp, "lib")]
    _run(cmd)
    return out


def build_all(force=False, verbose=False):
    import concurrent.futures as cf
    build_libgsx(force, verbose)
    with cf.ThreadPoolExecutor(2) as ex:  # the two shim builds are independent (each ~40 s of g++ over the ATen headers)
        for f in [ex.submit(build_ops_module, force), ex.submit(build_backend_lib, force)]:
            f.result()


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
    print("built", LIBGSX, "and", ops_module_path())
