"""A/B of the two forward blend kernels in one process (the switch GSX_FWD is read per launch): one list per 8x8 quadrant ("wave") vs
four lists per wave ("quad") vs two pixels per lane / eight lists per wave ("pair").  Outputs must be bit-identical; prints the medians of n launches.   python tools/fwd_quad_ab.py [1m|5m|dense] [n]"""
import os
os.environ.setdefault("GSX_TEST_SWITCHES", "1")
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import gsx  # noqa: E402,F401
from gsx import layout, ops, rasterizer, scenes  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "1m"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = "cuda:0"
FLUSH = torch.zeros(int(os.environ["GSX_AB_FLUSH"]) * 262144, device=dev) if os.environ.get("GSX_AB_FLUSH") else None
MODES = tuple(os.environ.get("GSX_AB_MODES", "wave,quad,pair").split(","))   # GSX_FWD values, the first one is the yardstick
for cam_kind in (("pinhole", "fisheye") if len(sys.argv) <= 3 else ("pinhole",)):
    list_tile = 16
    if which == "heavy":   # large footprints, long lists (the regime of a trained capture; lists per 32x32 pixels as the fused path picks there)
        scene = scenes.scene_frustum(500_000, 1296, 840, 1000.0, (2.0, 8.0), scale_range=(0.02, 0.12), sh_degree=0, seed=5)
        list_tile = 32
    elif which == "dense_hd":   # the saturated frame at 1920 x 1080: the same footprints in pixels and the same opacities, nine times the Gaussians
        scene = scenes.scene_frustum(2_700_000, 1920, 1080, 900.0, (2.0, 6.0), scale_range=(0.0033, 0.027), sh_degree=0, seed=3)
        scene["opacities"] = torch.rand(2_700_000, generator=torch.Generator().manual_seed(4)) * 0.3 + 0.69
    elif which == "dense":
        scene = scenes.scene_frustum(300_000, 640, 360, 300.0, (2.0, 6.0), scale_range=(0.01, 0.08), sh_degree=0, seed=3)
        scene["opacities"] = torch.rand(300_000, generator=torch.Generator().manual_seed(4)) * 0.3 + 0.69
    else:
        scene = {"1m": scenes.scene_1m, "5m": scenes.scene_5m}[which]()
    if which == "heavy":
        scene["opacities"] = torch.rand(500_000, generator=torch.Generator().manual_seed(4)) * 0.5 + 0.2
    perm = layout.morton_order(scene["means"])
    for k in ("means", "quats", "scales", "opacities", "sh"):
        scene[k] = scene[k][perm].contiguous()
    W, H = scene["width"], scene["height"]
    model = scenes.to_splat_data(scene, dev)
    fisheye = cam_kind == "fisheye"
    cam_model = ops.CameraModelType.FISHEYE if fisheye else ops.CameraModelType.PINHOLE
    radial = torch.tensor([0.01, -0.002, 0.0, 0.0]) if fisheye else None
    cam = rasterizer.Camera(viewmat=scene["viewmat"].to(dev), K=scene["K"].to(dev), width=W, height=H, camera_model=cam_model, radial=radial)
    with torch.no_grad():
        out = rasterizer.rasterize(cam, model, scene["background"].to(dev))
    radial = None if radial is None else radial.to(dev)
    d = lambda k: scene[k].to(dev)  # noqa: E731
    ut = ops.UnscentedTransformParameters()
    colors, off, fl = out.aux["colors"].contiguous(), out.aux["isect_offsets"], out.aux["flatten_ids"]
    if list_tile == 32:
        tw, th = (W + 31) // 32, (H + 31) // 32
        P = ops.projection_ut_3dgs_fused(d("means"), d("quats"), d("scales"), d("opacities"), d("viewmat")[None].contiguous(), None, d("K")[None].contiguous(), W, H, 0.3,
                                         0.01, 1e4, 0.0, False, cam_model, ut, ops.ShutterType.GLOBAL, radial, None, None)
        _, _, fl, off = ops.intersect_tile_binned(P[1], P[0], P[2], 1, 32, tw, th, False)
    common = (d("means"), d("quats"), d("scales"), colors, d("opacities")[None].contiguous(), d("background")[None].contiguous(), None, W, H, list_tile,
              d("viewmat")[None].contiguous(), None, d("K")[None].contiguous(), cam_model, ut, ops.ShutterType.GLOBAL, radial, None, None, off, fl)

    def timeit(fn):
        for _ in range(3):
            fn()
        ts = []
        for _ in range(n):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if FLUSH is not None:
                FLUSH.add_(1.0)   # GSX_AB_FLUSH=<MiB>: a streaming kernel over that many MiB in front of every timed launch (what the caches hold in a training step is not this kernel's own last run)
            s.record()
            r = fn()
            e.record()
            e.synchronize()
            ts.append(s.elapsed_time(e))
        ts.sort()
        return ts[len(ts) // 2], r

    res = {}
    for rnd in range(2):
        for mode in MODES:
            os.environ["GSX_FWD"] = mode
            t, r = timeit(lambda: ops.rasterize_to_pixels_from_world_3dgs_fwd(*common))
            res[mode] = [x.clone() for x in r[:3]]
            print(f"{which} {cam_kind} lists{list_tile} n_isects={fl.numel()} GSX_FWD={mode}: fwd op {t:.4f} ms")
    for mode in MODES[1:]:
        same = [bool(torch.equal(a, b)) for a, b in zip(res[MODES[0]], res[mode])]
        print("   %s vs %s: bit-identical renders / alphas / last_ids:" % (MODES[0], mode), same, " max |d rgb| %.3g" % float((res[MODES[0]][0] - res[mode][0]).abs().max()))
