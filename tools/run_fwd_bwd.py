"""Run the S-1M hot path a few times (for rocprofv3 --pmc / --kernel-trace runs): python tools/run_fwd_bwd.py [n] [fwd|all]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

if os.environ.get("GSX_VARIANT_LIB"):   # A/B: a variant build of libgsx.so preloaded under the same soname (tools/build_variant.sh): the extension binds to it
    import ctypes
    ctypes.CDLL(os.environ["GSX_VARIANT_LIB"], mode=ctypes.RTLD_GLOBAL)

import gsx  # noqa: E402,F401
from gsx import loss as gloss  # noqa: E402
from gsx import rasterizer, scenes  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
mode = sys.argv[2] if len(sys.argv) > 2 else "all"
dev = "cuda:0"
scene = {"1m": scenes.scene_1m, "5m": scenes.scene_5m}[os.environ.get("GSX_SCENE", "1m")]()
if os.environ.get("GSX_RANDOM_ORDER") != "1":   # bench.py's default memory order (gsx.layout)
    from gsx import layout
    order = layout.morton_order(scene["means"])
    for k in ("means", "quats", "scales", "opacities", "sh"):
        scene[k] = scene[k][order].contiguous()
model = scenes.to_splat_data(scene, dev)
cam = rasterizer.Camera(viewmat=scene["viewmat"].to(dev), K=scene["K"].to(dev), width=scene["width"], height=scene["height"])
bg = scene["background"].to(dev)
if mode in ("all", "unfused"):
    for p in model.params():
        p.requires_grad_(True)
target = torch.rand(3, scene["height"], scene["width"], generator=torch.Generator().manual_seed(1234)).to(dev)
for _ in range(n):
    if mode == "all":      # the bench step: fused render + fused L1/SSIM loss + backward
        out = rasterizer.rasterize_fused(cam, model, bg)
        gloss.photometric_loss(out.render_hwc, target, 0.2).backward()
    elif mode == "unfused":
        out = rasterizer.rasterize(cam, model, bg)
        out.image.sum().backward()
    else:
        with torch.no_grad():
            out = rasterizer.rasterize(cam, model, bg)
torch.cuda.synchronize()
print("n_isects", out.n_isects)
