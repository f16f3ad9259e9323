"""Does an HBM-bound per-Gaussian kernel overlap with the binned intersection's chain of small kernels when launched on a second stream?
(the question behind splitting the front end into geometry -> [intersection || SH colours + record colours] -> blend; round 6)
Serial: projection -> intersect_tile_binned_guarded -> sh_colors_fwd on one stream.  Overlapped: sh_colors_fwd on a side stream behind an event
recorded after the projection, joined before the end.  Same ops, same inputs; wall time over N iterations without a synchronisation in between."""
import sys
import time

import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import gsx  # noqa: F401
from gsx import ops, scenes, layout

DEV = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
sc = scenes.scene_1m()
order = layout.morton_order(sc["means"])
for k in ("means", "quats", "scales", "opacities", "sh"):
    sc[k] = sc[k][order].contiguous()
d = lambda k: sc[k].to(DEV).contiguous()  # noqa: E731
means, quats, scales, opac, sh = d("means"), d("quats"), d("scales"), d("opacities"), d("sh")
vm, K = sc["viewmat"][None].to(DEV).contiguous(), sc["K"][None].to(DEV).contiguous()
W, H = sc["width"], sc["height"]
tw, th = (W + 15) // 16, (H + 15) // 16
ut = ops.UnscentedTransformParameters()
side = torch.cuda.Stream()


def proj():
    return ops.projection_ut_3dgs_fused(means, quats, scales, opac, vm, None, K, W, H, 0.3, 0.01, 1e4, 0.0, False, ops.CameraModelType.PINHOLE, ut,
                                        ops.ShutterType.GLOBAL, None, None, None)


def serial():
    radii, m2d, dep, _, _ = proj()
    ops.intersect_tile_binned_guarded(m2d, radii, dep, 1, 16, tw, th)
    ops.sh_colors_fwd(3, means, vm, sh, radii)


def overlapped():
    radii, m2d, dep, _, _ = proj()
    ev = torch.cuda.Event()
    ev.record()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        col = ops.sh_colors_fwd(3, means, vm, sh, radii)
        ev2 = torch.cuda.Event()
        ev2.record()
    ops.intersect_tile_binned_guarded(m2d, radii, dep, 1, 16, tw, th)
    torch.cuda.current_stream().wait_event(ev2)
    return col


def only(which):
    radii, m2d, dep, _, _ = proj()
    if which == "isect":
        ops.intersect_tile_binned_guarded(m2d, radii, dep, 1, 16, tw, th)
    elif which == "sh":
        ops.sh_colors_fwd(3, means, vm, sh, radii)


def run(fn, *a):
    for _ in range(20):
        fn(*a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn(*a)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for rnd in range(3):
    print("round %d: projection only %.1f us | + intersection %.1f | + SH colours %.1f | serial (all three) %.1f | SH colours on a side stream %.1f"
          % (rnd, run(only, "none"), run(only, "isect"), run(only, "sh"), run(serial), run(overlapped)))
