"""Binned intersect pipeline vs the device-wide-sort pipeline: bit-exact comparison and timing on S-1M (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gsx  # noqa: F401
from gsx import ops, rasterizer, scenes

def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

dev = "cuda:0"
for name in (sys.argv[1:] or ["1m"]):
    scene = {"small": scenes.scene_small, "1m": scenes.scene_1m, "5m": scenes.scene_5m}[name]()
    model = scenes.to_splat_data(scene, dev)
    W, H = scene["width"], scene["height"]
    cam = rasterizer.Camera(viewmat=scene["viewmat"].to(dev), K=scene["K"].to(dev), width=W, height=H)
    with torch.no_grad():
        scales, quats, opac = ops.splat_activations_fwd(model.scaling_raw, model.rotation_raw, model.opacity_raw.reshape(-1))
        ut = ops.UnscentedTransformParameters()
        radii, means2d, depths, conics, _ = ops.projection_ut_3dgs_fused(
            model.means, quats, scales, opac, cam.world_view_transform().contiguous(), None, cam.K_batched().contiguous(), W, H, 0.3, 0.01, 1e10, 0.0,
            False, ops.CameraModelType.PINHOLE, ut, ops.ShutterType.GLOBAL, None, None, None)
    tw, th = (W + 15) // 16, (H + 15) // 16
    tpg, ids, flat = ops.intersect_tile_device_sort(means2d, radii, depths, 1, 16, tw, th, True)
    off = ops.intersect_offset(ids, 1, tw, th)
    tpg2, ids2, flat2, off2 = ops.intersect_tile_binned(means2d, radii, depths, 1, 16, tw, th, True)
    print(name, "n_isects", flat.numel(), flat2.numel(), "equal: tpg", torch.equal(tpg, tpg2), "flatten", torch.equal(flat, flat2),
          "isect_ids", torch.equal(ids, ids2), "offsets", torch.equal(off, off2))
    print("  device-wide sort + offsets: %.3f ms" % timeit(lambda: ops.intersect_offset(ops.intersect_tile_device_sort(means2d, radii, depths, 1, 16, tw, th, True)[1], 1, tw, th)))
    print("  binned (with isect_ids)   : %.3f ms" % timeit(lambda: ops.intersect_tile_binned(means2d, radii, depths, 1, 16, tw, th, True)))
    print("  binned (flatten only)     : %.3f ms" % timeit(lambda: ops.intersect_tile_binned(means2d, radii, depths, 1, 16, tw, th, False)))
