"""Times the two blend ops in isolation (HIP events, median of n launches) on S-1M or S-5M: the A/B harness for kernel variants
selected by environment switches (GSX_BWD=pm = pixel-major backward, GSX_RASTER_PATH=generic, GSX_AB_CAMERA=fisheye|rolling), one process per variant;
GSX_AB_SAVE=path keeps the gradients for tools/blend_ab_compare.py.   python tools/blend_ab.py [1m|5m|dense] [n]"""
import os
os.environ.setdefault("GSX_TEST_SWITCHES", "1")   # this tool flips libgsx's A/B switches (include/gsx.h: gsx_test_switch)
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

if os.environ.get("GSX_VARIANT_LIB"):   # a variant build of libgsx.so preloaded under the same soname (tools/build_variant.sh): the extension binds to it
    import ctypes
    ctypes.CDLL(os.environ["GSX_VARIANT_LIB"], mode=ctypes.RTLD_GLOBAL)

import gsx  # noqa: E402,F401
from gsx import ops, rasterizer, scenes  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "1m"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = "cuda:0"
if which == "dense":   # saturated tiles: high opacity, large footprints, early termination everywhere (the regime of a trained real capture)
    scene = scenes.scene_frustum(300_000, 640, 360, 300.0, (2.0, 6.0), scale_range=(0.01, 0.08), sh_degree=0, seed=3)
    scene["opacities"] = torch.rand(300_000, generator=torch.Generator().manual_seed(4)) * 0.3 + 0.69
else:
    scene = {"1m": scenes.scene_1m, "5m": scenes.scene_5m}[which]()
W, H = scene["width"], scene["height"]
model = scenes.to_splat_data(scene, dev)
fisheye = os.environ.get("GSX_AB_CAMERA") == "fisheye"   # the same scene through an equidistant fisheye of the same focal length
cam_model = ops.CameraModelType.FISHEYE if fisheye else ops.CameraModelType.PINHOLE
radial = torch.tensor([0.01, -0.002, 0.0, 0.0]) if fisheye else None
cam = rasterizer.Camera(viewmat=scene["viewmat"].to(dev), K=scene["K"].to(dev), width=W, height=H, camera_model=cam_model, radial=radial)
with torch.no_grad():
    out = rasterizer.rasterize(cam, model, scene["background"].to(dev))
radial = None if radial is None else radial.to(dev)
rolling = os.environ.get("GSX_AB_CAMERA") == "rolling"   # timing only: the global-shutter binning, blended with a rolling shutter
shutter = ops.ShutterType.ROLLING_TOP_TO_BOTTOM if rolling else ops.ShutterType.GLOBAL
vm1 = None
if rolling:
    vm1 = scene["viewmat"].clone()
    vm1[0, 3], vm1[1, 3] = 0.01, -0.01
    vm1 = vm1[None].to(dev).contiguous()
d = lambda k: scene[k].to(dev)  # noqa: E731
ut = ops.UnscentedTransformParameters()
colors, off, fl = out.aux["colors"].contiguous(), out.aux["isect_offsets"], out.aux["flatten_ids"]
common = (d("means"), d("quats"), d("scales"), colors, d("opacities")[None].contiguous(), d("background")[None].contiguous(), None, W, H, 16,
          d("viewmat")[None].contiguous(), vm1, d("K")[None].contiguous(), cam_model, ut, shutter, radial, None, None, off, fl)
g = torch.Generator(device=dev).manual_seed(0)
v_rc, v_ra = torch.randn(1, H, W, 3, device=dev, generator=g), torch.randn(1, H, W, 1, device=dev, generator=g)


def timeit(fn):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = fn()
        e.record()
        e.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    return ts[len(ts) // 2], r


t_f, fwd = timeit(lambda: ops.rasterize_to_pixels_from_world_3dgs_fwd(*common, keep_ws=True))
t_b, bwd = timeit(lambda: ops.rasterize_to_pixels_from_world_3dgs_bwd(*common, fwd[1], fwd[2], v_rc, v_ra, fwd_ws=fwd[3]))
chk = [float(x.double().abs().sum()) for x in bwd]
print("%s%s GSX_BWD=%s GSX_RASTER_PATH=%s n_isects=%d  fwd %.4f ms  bwd %.4f ms  |grads|_1 = %s" % (which, " fisheye" if fisheye else (" rolling-shutter" if rolling else ""), os.environ.get("GSX_BWD", "-"), os.environ.get("GSX_RASTER_PATH", "-"), fl.numel(), t_f, t_b,
                                                                              " ".join("%.6g" % c for c in chk)))
if os.environ.get("GSX_AB_SAVE"):   # gradients of this variant, for tools/blend_ab_compare.py
    torch.save([x.cpu() for x in bwd], os.environ["GSX_AB_SAVE"])
