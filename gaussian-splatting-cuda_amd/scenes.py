"""Synthetic scenes of BASELINE.json's configs (exact distributions: SURVEY.md §8d).  All tensors are
generated on the CPU with torch.manual_seed-style generators so the CPU oracle and the GPU see identical bits."""
import math

import torch


def _gen(seed):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return g


def look_at_viewmat(eye, target, up=(0.0, -1.0, 0.0)):
    """Row-major world->camera [4,4] with +z forward (OpenCV convention, as the reference's cameras)."""
    eye = torch.tensor(eye, dtype=torch.float64)
    target = torch.tensor(target, dtype=torch.float64)
    upv = torch.tensor(up, dtype=torch.float64)
    z = target - eye
    z = z / z.norm()
    x = torch.linalg.cross(-upv, z)
    x = x / x.norm()
    y = torch.linalg.cross(z, x)
    R = torch.stack([x, y, z], 0)
    t = -R @ eye
    vm = torch.eye(4, dtype=torch.float64)
    vm[:3, :3] = R
    vm[:3, 3] = t
    return vm.float()


def intrinsics(fx, fy, cx, cy):
    K = torch.zeros(3, 3)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2], K[2, 2] = fx, fy, cx, cy, 1.0
    return K


def scene_small(seed=42, N=10_000):
    """cfg1 / S-small: 10k Gaussians, SH degree 0, identity camera, 256x256, fx=fy=200 (tests/test_rasterization.cpp:144-163)."""
    g = _gen(seed)
    xy = torch.rand(N, 2, generator=g) * 2 - 1
    z = 2 + (torch.rand(N, 1, generator=g) * 2 - 1).abs()
    means = torch.cat([xy, z], 1)
    quats = torch.nn.functional.normalize(torch.randn(N, 4, generator=g), dim=-1)
    scales = torch.rand(N, 3, generator=g) * 0.05 + 0.01
    opacities = torch.rand(N, generator=g) * 0.5 + 0.3
    sh = (torch.rand(N, 1, 3, generator=g) - 0.5) * 0.3
    return dict(means=means, quats=quats, scales=scales, opacities=opacities, sh=sh, sh_degree=0,
                viewmat=torch.eye(4), K=intrinsics(200.0, 200.0, 128.0, 128.0), width=256, height=256,
                background=torch.zeros(3))


def scene_frustum(N, width, height, focal, z_range, scale_range=(0.002, 0.02), sh_degree=3, seed=42,
                  background=(0.1, 0.1, 0.1)):
    """S-1M / S-5M family: Gaussians uniform in the camera frustum slab, log-uniform scales."""
    g = _gen(seed)
    z = torch.rand(N, generator=g) * (z_range[1] - z_range[0]) + z_range[0]
    hx, hy = 0.5 * width / focal, 0.5 * height / focal
    x = (torch.rand(N, generator=g) * 2 - 1) * hx * z
    y = (torch.rand(N, generator=g) * 2 - 1) * hy * z
    means = torch.stack([x, y, z], 1)
    quats = torch.nn.functional.normalize(torch.randn(N, 4, generator=g), dim=-1)
    lo, hi = math.log(scale_range[0]), math.log(scale_range[1])
    scales = torch.exp(torch.rand(N, 3, generator=g) * (hi - lo) + lo)
    opacities = torch.rand(N, generator=g) * 0.8 + 0.1
    K = (sh_degree + 1) ** 2
    sh = (torch.rand(N, K, 3, generator=g) - 0.5) * 0.3
    return dict(means=means, quats=quats, scales=scales, opacities=opacities, sh=sh, sh_degree=sh_degree,
                viewmat=torch.eye(4), K=intrinsics(focal, focal, width / 2.0, height / 2.0), width=width,
                height=height, background=torch.tensor(background))


def scene_1m(seed=42, N=1_000_000):
    """cfg2 / S-1M: 1M Gaussians, SH degree 3, 1920x1080, fx=fy=1000."""
    return scene_frustum(N, 1920, 1080, 1000.0, (2.0, 10.0), seed=seed)


def scene_5m(seed=42, N=5_000_000):
    """cfg5 / S-5M: 5M Gaussians, 3840x2160, fx=fy=2000, z in U(2,14)."""
    return scene_frustum(N, 3840, 2160, 2000.0, (2.0, 14.0), seed=seed)


def ring_cameras(n, radius=6.0, target=(0.0, 0.0, 6.0), height=0.0):
    """S-8cam: n cameras on a ring around the slab centre, looking at it."""
    cams = []
    for i in range(n):
        a = 2 * math.pi * i / n
        eye = (target[0] + radius * math.sin(a), target[1] + height, target[2] - radius * math.cos(a))
        cams.append(look_at_viewmat(eye, target))
    return cams


def to_splat_data(scene, device):
    """Raw (pre-activation) parameters whose activations reproduce the scene tensors (splat_data.cpp:267-286)."""
    from .rasterizer import SplatData
    sh = scene["sh"].to(device)
    op = scene["opacities"].to(device).clamp(1e-6, 1 - 1e-6)
    return SplatData(means=scene["means"].to(device).clone(), sh=sh.contiguous().clone(),
                     scaling_raw=torch.log(scene["scales"].to(device)),
                     rotation_raw=scene["quats"].to(device).clone(), opacity_raw=torch.logit(op).unsqueeze(-1),
                     active_sh_degree=scene["sh_degree"])
