"""The drop-in boundary, compiled the way the reference would use it: tests/compat/ref_call_sites.cpp includes the backend headers
by the reference's file names ("Ops.h", "Projection.h", "Common.h" through -I compat/gsplat) next to <torch/torch.h>, calls all ten
`gsplat::` operators with the argument shapes of the reference call sites, and links libgsx_gsplat_backend.so + libgsx.so (no Python
in that process).  CPU: the TU compiles, links and its host-side checks pass.  GPU: the whole chain runs and matches the Python ops."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gaussian-splatting-cuda_amd")
SRC = os.path.join(ROOT, "tests", "compat", "ref_call_sites.cpp")
EXE = os.path.join(ROOT, "tests", "compat", "ref_call_sites")


def _build():
    sys.path.insert(0, PKG)
    try:
        import build as gbuild  # gaussian-splatting-cuda_amd/build.py
    finally:
        sys.path.pop(0)
    gbuild.build_all()
    deps = [SRC, gbuild.BACKEND] + [os.path.join(ROOT, "compat", "gsplat", h) for h in ("Ops.h", "Cameras.h", "Common.h", "Projection.h")]
    if os.path.exists(EXE) and all(os.path.getmtime(d) <= os.path.getmtime(EXE) for d in deps):
        return EXE
    inc, link = gbuild.torch_cxx_flags()
    cmd = ["g++", "-O1", "-I" + os.path.join(ROOT, "compat", "gsplat")] + inc + [SRC, "-o", EXE, "-L" + PKG, "-lgsx_gsplat_backend", "-lgsx"] + link + \
          ["-Wl,-rpath," + PKG]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-4000:]
    return EXE


def test_reference_call_sites_compile_link_and_host_checks():
    exe = _build()
    r = subprocess.run([exe, "cpu"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0 and "CPU OK" in r.stdout, r.stdout[-2000:]
    # no Python, no pybind in the process: the executable does not depend on libpython / libtorch_python
    ldd = subprocess.run(["ldd", exe], stdout=subprocess.PIPE, text=True).stdout
    libs = [ln.split()[0] for ln in ldd.splitlines() if ln.strip()]
    assert "libgsx_gsplat_backend.so" in libs and "libgsx.so" in libs
    assert not any(n.startswith("libpython") or n.startswith("libtorch_python") for n in libs), libs


def test_compat_headers_define_each_type_once():
    """Cameras.h / Common.h are the single definition of the three ABI types: gsx_ops.h and Ops.h only include them."""
    seen = {}
    for h in ["compat/gsplat/Ops.h", "compat/gsplat/Cameras.h", "compat/gsplat/Common.h", "compat/gsplat/Projection.h", "include/gsx_ops.h",
              "include/gsx_training_ops.h"]:
        src = open(os.path.join(ROOT, h)).read()
        for name in re.findall(r"^\s*(?:enum(?:\s+class)?|struct)\s+(CameraModelType|ShutterType|UnscentedTransformParameters)\b[^;]*\{", src, re.M):
            seen.setdefault(name, []).append(h)
    assert seen == {"CameraModelType": ["compat/gsplat/Common.h"], "ShutterType": ["compat/gsplat/Cameras.h"],
                    "UnscentedTransformParameters": ["compat/gsplat/Cameras.h"]}, seen


@pytest.mark.gpu
def test_reference_call_sites_run_on_gpu_and_match_python_ops():
    import torch
    import gsx  # noqa: F401
    from gsx import ops
    exe = _build()
    r = subprocess.run([exe, "gpu"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "GPU OK" in r.stdout, r.stdout[-2000:]
    got = {k: float(v) for k, v in re.findall(r"(\w+)=([-+0-9.e]+)", r.stdout)}
    # the same closed-form scene through the Python bindings of the same shim
    dev, N, W, H = "cuda:0", 3000, 160, 96
    i = torch.arange(N, dtype=torch.float64)
    x, y, z = torch.sin(i * 12.9898) * 0.9, torch.cos(i * 78.233) * 0.9, 2.0 + torch.frac(i * 0.61803398875) * 2.0
    means = torch.stack([x * z * 0.5, y * z * 0.5, z], 1).float().to(dev)
    quats = torch.stack([torch.cos(i * 0.37) + 1.5, torch.sin(i * 1.1), torch.cos(i * 2.3), torch.sin(i * 0.7)], 1).float().to(dev)
    scales = (0.01 + 0.04 * torch.stack([torch.frac(i * 0.137), torch.frac(i * 0.731), torch.frac(i * 0.377)], 1)).float().to(dev)
    opac = (0.3 + 0.6 * torch.frac(i * 0.2718)).float().to(dev)
    k = torch.arange(N * 16 * 3, dtype=torch.float64).reshape(N, 16, 3)
    sh = (0.3 * (torch.frac(k * 0.0123457) - 0.5)).float().to(dev)
    vm = torch.eye(4, device=dev).unsqueeze(0)
    K = torch.tensor([[0.8 * W, 0, 0.5 * W], [0, 0.8 * W, 0.5 * H], [0, 0, 1]], dtype=torch.float32, device=dev).unsqueeze(0)
    ut = ops.UnscentedTransformParameters()
    radii, means2d, depths, _, _ = ops.projection_ut_3dgs_fused(means, quats, scales, opac, vm, None, K, W, H, 0.3, 0.01, 1e4, 0.0, False,
                                                               ops.CameraModelType.PINHOLE, ut, ops.ShutterType.GLOBAL, None, None, None)
    masks = (radii > 0).all(-1)
    colors = ops.spherical_harmonics_fwd(3, means.contiguous(), sh, masks.reshape(-1)).reshape(1, N, 3)
    colors = (torch.clamp_min(colors + 0.5, 0.0) * masks.unsqueeze(-1)).contiguous()
    tw, th = (W + 15) // 16, (H + 15) // 16
    _, ids, fl = ops.intersect_tile(means2d, radii, depths, None, None, 1, 16, tw, th, True)
    off = ops.intersect_offset(ids, 1, tw, th).reshape(1, th, tw)
    bg = torch.full((1, 3), 0.1, device=dev)
    ren, alp, last = ops.rasterize_to_pixels_from_world_3dgs_fwd(means, quats, scales, colors, opac[None].contiguous(), bg, None, W, H, 16, vm, None, K,
                                                                 ops.CameraModelType.PINHOLE, ut, ops.ShutterType.GLOBAL, None, None, None, off, fl)
    assert got["n_isects"] == fl.numel()
    assert abs(got["render"] - float(ren.double().sum())) <= 1e-6 * abs(got["render"])
    assert abs(got["alpha"] - float(alp.double().sum())) <= 1e-6 * abs(got["alpha"])
    g = ops.rasterize_to_pixels_from_world_3dgs_bwd(means, quats, scales, colors, opac[None].contiguous(), bg, None, W, H, 16, vm, None, K,
                                                    ops.CameraModelType.PINHOLE, ut, ops.ShutterType.GLOBAL, None, None, None, off, fl, alp, last,
                                                    torch.full_like(ren, 0.5), torch.ones_like(alp))
    # the backward's summation order is scheduling dependent: agreement to rounding, not bit for bit
    assert abs(got["v_means_abs"] - float(g[0].abs().double().sum())) <= 1e-4 * got["v_means_abs"]
    assert abs(got["v_scales_abs"] - float(g[2].abs().double().sum())) <= 1e-4 * got["v_scales_abs"]
    assert abs(got["rot"] - float(ops.quats_to_rotmats(quats).abs().double().sum())) <= 1e-6 * got["rot"]
