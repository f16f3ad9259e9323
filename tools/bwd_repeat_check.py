"""Repeatability of the blend backward (the Gaussian-major kernel's locked LDS flush and the chained records make the summation order depend
on scheduling): n launches on the same inputs must agree to rounding.  A race (lost update) would show as a ~1e-4..1e-2 outlier.
python tools/bwd_repeat_check.py [1m|5m|dense] [n]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import gsx  # noqa: E402,F401
from gsx import ops, rasterizer, scenes  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "1m"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = "cuda:0"
if which == "dense":   # saturated tiles: high opacity, large footprints, early termination everywhere
    scene = scenes.scene_frustum(300_000, 640, 360, 300.0, (2.0, 6.0), scale_range=(0.01, 0.08), sh_degree=0, seed=3)
    scene["opacities"] = torch.rand(300_000, generator=torch.Generator().manual_seed(4)) * 0.3 + 0.69
else:
    scene = {"1m": scenes.scene_1m, "5m": scenes.scene_5m}[which]()
W, H = scene["width"], scene["height"]
model = scenes.to_splat_data(scene, dev)
cam = rasterizer.Camera(viewmat=scene["viewmat"].to(dev), K=scene["K"].to(dev), width=W, height=H)
with torch.no_grad():
    out = rasterizer.rasterize(cam, model, scene["background"].to(dev))
d = lambda k: scene[k].to(dev)  # noqa: E731
ut = ops.UnscentedTransformParameters()
colors, off, fl = out.aux["colors"].contiguous(), out.aux["isect_offsets"], out.aux["flatten_ids"]
common = (d("means"), d("quats"), d("scales"), colors, d("opacities")[None].contiguous(), d("background")[None].contiguous(), None, W, H, 16,
          d("viewmat")[None].contiguous(), None, d("K")[None].contiguous(), ops.CameraModelType.PINHOLE, ut, ops.ShutterType.GLOBAL, None, None, None, off, fl)
g = torch.Generator(device=dev).manual_seed(0)
v_rc, v_ra = torch.randn(1, H, W, 3, device=dev, generator=g), torch.randn(1, H, W, 1, device=dev, generator=g)
fwd = ops.rasterize_to_pixels_from_world_3dgs_fwd(*common, keep_ws=True)
ref = [x.double() for x in ops.rasterize_to_pixels_from_world_3dgs_bwd(*common, fwd[1], fwd[2], v_rc, v_ra, fwd_ws=fwd[3])]
worst = [0.0] * 5
for _ in range(n):
    cur = ops.rasterize_to_pixels_from_world_3dgs_bwd(*common, fwd[1], fwd[2], v_rc, v_ra, fwd_ws=fwd[3])
    for i, (a, b) in enumerate(zip(ref, cur)):
        worst[i] = max(worst[i], float((a - b.double()).norm() / a.norm().clamp_min(1e-300)))
print("%s n_isects=%d  %d repeats  worst rel-L2 vs the first launch: means %.1e quats %.1e scales %.1e colours %.1e opacities %.1e  alpha mean %.3f" %
      ((which, fl.numel(), n) + tuple(worst) + (float(fwd[1].mean()),)))
assert max(worst) < 1e-5, "backward launches disagree beyond rounding"
