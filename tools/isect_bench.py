"""Micro-benchmark of the binned intersection at garden shape (1296x840, ring camera, central cluster): builds the garden stand-in's
hidden scene at N Gaussians, projects it for one camera and times ops.intersect_tile_binned alone; prints the segment histogram.
With train=K the model is the stand-in's trainee after K training iterations instead (MCMC-grown: much fatter tiles than the hidden scene).
Run on the GPU box:  python tools/isect_bench.py [N=1000000] [scale_mul=1.0] [reps=30] [train=0]"""
import importlib.util
import os
os.environ.setdefault("GSX_TEST_SWITCHES", "1")   # this tool flips libgsx's A/B switches (include/gsx.h: gsx_test_switch)
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import gsx  # noqa: E402,F401
from gsx import ops, rasterizer  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    mul = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    spec = importlib.util.spec_from_file_location("garden", os.path.join(ROOT, "examples", "train_garden_standin.py"))
    garden = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(garden)
    train = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    dev = "cuda:0"
    if train:
        tr, model, cams, _, _, _ = garden.setup(dev)
        for it in range(1, train + 1):
            tr.train_step(it)
        cam, n = cams[4], model.means.shape[0]
    else:
        garden.N_GT = n
        model = garden.hidden_scene(dev)
        cam = garden.ring_cameras(dev)[4]
    if os.environ.get("GSX_ISECT_SHUFFLE") == "1":   # the same model in random memory order (layout.py keeps the trainee in Morton order)
        perm = torch.randperm(model.means.shape[0], device=dev)
        for name in ("means", "sh", "scaling_raw", "rotation_raw", "opacity_raw"):
            setattr(model, name, getattr(model, name).data.index_select(0, perm))
    ut = ops.UnscentedTransformParameters()
    with torch.no_grad():
        scales, quats, opac, radii, means2d, depths, conics = ops.splat_activations_projection_ut(
            model.means.contiguous(), (model.scaling_raw + torch.log(torch.tensor(mul))).contiguous(), model.rotation_raw.contiguous(),
            model.opacity_raw.reshape(-1).contiguous(), cam.viewmat[None], cam.K[None], cam.width, cam.height, rasterizer.EPS2D,
            rasterizer.NEAR_PLANE, rasterizer.FAR_PLANE, rasterizer.RADIUS_CLIP, ops.CameraModelType.PINHOLE, ut, None, None, None)
    tw, th = (cam.width + 15) // 16, (cam.height + 15) // 16
    for fill in ("keys", "ranked"):
        os.environ["GSX_INTERSECT_FILL"] = fill
        for _ in range(3):
            _, _, flat, off = ops.intersect_tile_binned(means2d, radii, depths, 1, 16, tw, th, False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.intersect_tile_binned(means2d, radii, depths, 1, 16, tw, th, False)
        e1.record()
        torch.cuda.synchronize()
        print("fill=%-6s N %d  visible %d  n_isects %d  tiles %d  intersect_tile_binned %.3f ms" % (
            fill, n, int((radii > 0).all(-1).sum()), flat.numel(), off.numel(), e0.elapsed_time(e1) / reps))
    seg = torch.cat([off.flatten(), torch.tensor([flat.numel()], device=dev, dtype=off.dtype)])
    seg = (seg[1:] - seg[:-1]).cpu()
    edges = [0, 1, 1025, 4097, 8193, 16385, 32769, 65537, 1 << 30]
    for lo, hi in zip(edges[:-1], edges[1:]):
        m = (seg >= lo) & (seg < hi)
        print("  segments with %6d <= keys < %10d: %5d  (%9d keys)" % (lo, hi, int(m.sum()), int(seg[m].sum())))
    print("  largest segment:", int(seg.max()))


if __name__ == "__main__":
    main()
