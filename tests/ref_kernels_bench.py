"""The reference's OWN kernels (its .cu files compiled unmodified for gfx950: oracle/_ref/gsplat_ref_hip{,_fast}.so, gsplat_ref_train.so)
timed beside this library's operators on the same MI355X, same tensors, same operator interface (gsplat/Ops.h) — S-1M @1080p.
Checker libraries only: nothing here is the product path (it lives under tests/ because only tests may load oracle/).  Prints a markdown table.
Usage (GPU box): python tests/ref_kernels_bench.py [morton|generator] > gpurun_out/ref_kernels.md"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib  # noqa: E402

gsx = importlib.import_module("gaussian-splatting-cuda_amd")
sys.modules.setdefault("gsx", gsx)
from gsx import layout, loss, ops, scenes  # noqa: E402
from oracle import ref_hip  # noqa: E402

DEV = "cuda:0"
order = sys.argv[1] if len(sys.argv) > 1 else "morton"


def timed(fn, reps=10, warm=3):
    for _ in range(warm):
        out = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


sc = scenes.scene_1m()
if order == "morton":
    perm = layout.morton_order(sc["means"])
    for k in ("means", "quats", "scales", "opacities", "sh"):
        sc[k] = sc[k][perm].contiguous()
d = lambda t: t.to(DEV).contiguous()  # noqa: E731
means, quats, scales, opac, sh = d(sc["means"]), d(sc["quats"]), d(sc["scales"]), d(sc["opacities"]), d(sc["sh"])
vm, K, bg = d(sc["viewmat"][None]), d(sc["K"][None]), d(sc["background"][None])
W, H, deg = sc["width"], sc["height"], sc["sh_degree"]
tw, th = (W + 15) // 16, (H + 15) // 16
N = means.shape[0]
rng = np.random.default_rng(3)
v_rc = torch.from_numpy(rng.standard_normal((1, H, W, 3)).astype(np.float32)).to(DEV)
v_ra = torch.from_numpy(rng.standard_normal((1, H, W, 1)).astype(np.float32)).to(DEV)
gt = torch.from_numpy(rng.random((1, 3, H, W), dtype=np.float32)).to(DEV)
rows = []


def reference(tag, ref):
    t = {}
    t["projection_ut_3dgs_fused"], P = timed(lambda: ref.projection_ut_3dgs_fused(means, quats, scales, opac, vm, None, K, W, H, 0.3, 0.01, 1e4, 0.0, False, 0, None, 4,
                                                                                   None, None, None))
    radii, means2d, depths = P[0], P[1], P[2]
    campos = torch.linalg.inv(vm.double())[:, :3, 3].float()
    dirs = (means[None] - campos[:, None]).contiguous()
    masks = (radii > 0).all(-1)
    shc = sh[None].contiguous()
    t["spherical_harmonics_fwd"], col = timed(lambda: ref.spherical_harmonics_fwd(deg, dirs, shc, masks))
    colors = torch.clamp_min(col + 0.5, 0.0)
    t["intersect_tile (+ device sort)"], I = timed(lambda: ref.intersect_tile(means2d, radii, depths, None, None, 1, 16, tw, th, True))
    t["intersect_offset"], off = timed(lambda: ref.intersect_offset(I[1], 1, tw, th))
    op = opac[None].contiguous()
    fa = (means, quats, scales, colors, op, bg, None, W, H, 16, vm, None, K, 0, None, 4, None, None, None, off, I[2])
    t["rasterize_to_pixels_from_world_3dgs_fwd"], F = timed(lambda: ref.rasterize_to_pixels_from_world_3dgs_fwd(*fa))
    t["rasterize_to_pixels_from_world_3dgs_bwd"], B = timed(lambda: ref.rasterize_to_pixels_from_world_3dgs_bwd(*fa, F[1], F[2], v_rc, v_ra))
    t["spherical_harmonics_bwd"], _ = timed(lambda: ref.spherical_harmonics_bwd(shc.shape[-2], deg, dirs, shc, masks, B[3], True))
    return t, int(I[2].numel())


def ours():
    t = {}
    ut = ops.UnscentedTransformParameters()
    cm, shut = ops.CameraModelType.PINHOLE, ops.ShutterType.GLOBAL
    t["projection_ut_3dgs_fused"], P = timed(lambda: ops.projection_ut_3dgs_fused(means, quats, scales, opac, vm, None, K, W, H, 0.3, 0.01, 1e4, 0.0, False, cm, ut, shut,
                                                                                   None, None, None))
    radii, means2d, depths = P[0], P[1], P[2]
    campos = torch.linalg.inv(vm.double())[:, :3, 3].float()
    dirs = (means[None] - campos[:, None]).contiguous()
    masks = (radii > 0).all(-1)
    shc = sh[None].contiguous()
    t["spherical_harmonics_fwd"], col = timed(lambda: ops.spherical_harmonics_fwd(deg, dirs, shc, masks))
    colors = torch.clamp_min(col + 0.5, 0.0)
    t["intersect_tile (+ device sort)"], I = timed(lambda: ops.intersect_tile_binned(means2d, radii, depths, 1, 16, tw, th, True))
    t["intersect_offset"] = 0.0   # the binned intersection returns the offsets
    off = I[3]
    op = opac[None].contiguous()
    fa = (means, quats, scales, colors, op, bg, None, W, H, 16, vm, None, K, cm, ut, shut, None, None, None, off, I[2])
    t["rasterize_to_pixels_from_world_3dgs_fwd"], F = timed(lambda: ops.rasterize_to_pixels_from_world_3dgs_fwd(*fa))
    t["rasterize_to_pixels_from_world_3dgs_bwd"], B = timed(lambda: ops.rasterize_to_pixels_from_world_3dgs_bwd(*fa, F[1], F[2], v_rc, v_ra))
    t["spherical_harmonics_bwd"], _ = timed(lambda: ops.spherical_harmonics_bwd(shc.shape[-2], deg, dirs, shc, masks, B[3], True))
    return t, F


def loss_ref(train, img_hwc):
    def run():
        r = img_hwc.detach().clone().requires_grad_(True)
        rendered = r.clamp(0, 1).permute(0, 3, 1, 2)
        val = 0.8 * torch.nn.functional.l1_loss(rendered, gt) + 0.2 * (1.0 - train.fused_ssim(rendered, gt, "valid", True))
        val.backward()
        return r.grad
    return timed(run)[0]


def loss_ours(img_hwc):
    def run():
        r = img_hwc.detach().clone().requires_grad_(True)
        loss.photometric_loss(r, gt, 0.2).backward()
        return r.grad
    return timed(run)[0]


def adam_all(step_fn):
    """one Adam step over the 59 floats of every Gaussian, group by group as fused_adam.cpp does (means, sh0, shN, scaling, rotation, opacity)"""
    groups = [torch.randn(N * k, device=DEV) for k in (3, 3, 45, 3, 4, 1)]
    state = [(torch.zeros_like(g), torch.zeros_like(g), torch.randn_like(g)) for g in groups]
    def run():
        for p, (m, v, g) in zip(groups, state):
            step_fn(p, m, v, g, 1e-3, 0.9, 0.999, 1e-15, 10.0, 31.6)
    return timed(run)[0]


mine, F = ours()
ref_ieee, n_isects = reference("IEEE", ref_hip.load(False))
ref_fast, _ = reference("fast", ref_hip.load(True))
train = ref_hip.load_train()
img = F[0]
mine["photometric loss fwd + bwd (L1 + fused SSIM, autograd)"] = loss_ours(img)
ref_ieee["photometric loss fwd + bwd (L1 + fused SSIM, autograd)"] = ref_fast["photometric loss fwd + bwd (L1 + fused SSIM, autograd)"] = loss_ref(train, img)
mine["Adam step, 59 floats x N (6 groups)"] = adam_all(ops.adam_step)
ref_ieee["Adam step, 59 floats x N (6 groups)"] = ref_fast["Adam step, 59 floats x N (6 groups)"] = adam_all(train.adam_step)
print(f"# Reference kernels vs gsx operators on one MI355X — S-1M @1080p ({N} Gaussians, {n_isects} intersections, {order} memory order)\n")
print("Reference = its own `.cu` files compiled unmodified for gfx950 (`oracle/build_ref_hip.sh`, `build_ref_train.sh`): IEEE flags and its release flag")
print("`--use_fast_math`.  gsx = this library behind the same `gsplat/Ops.h` operator interface, one call per operator (the training step of `bench.py` fuses")
print("further: projection + SH + records in one kernel, SH backward + Adam in one kernel).  ms per call, HIP events, 10 calls after 3 warm-up calls.\n")
print("| operator | reference (IEEE) | reference (fast-math) | gsx | gsx vs reference (fast-math) |")
print("|---|---|---|---|---|")
tot = [0.0, 0.0, 0.0]
for k in mine:
    a, b, c = ref_ieee[k], ref_fast[k], mine[k]
    tot[0] += a; tot[1] += b; tot[2] += c
    print(f"| `{k}` | {a:.3f} | {b:.3f} | {c:.3f} | {b / c:.1f}x |" if c > 0 else f"| `{k}` | {a:.3f} | {b:.3f} | (in intersect_tile) | |")
print(f"| **sum** | **{tot[0]:.3f}** | **{tot[1]:.3f}** | **{tot[2]:.3f}** | **{tot[1] / tot[2]:.1f}x** |")
