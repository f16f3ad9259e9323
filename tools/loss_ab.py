"""A/B timing of the fused photometric loss kernels (HIP events, 1080p unless H W are given): python tools/loss_ab.py [H W]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if os.environ.get("GSX_VARIANT_LIB"):   # a variant build of libgsx.so preloaded under the same soname: the extension binds to it
    import ctypes
    ctypes.CDLL(os.environ["GSX_VARIANT_LIB"], mode=ctypes.RTLD_GLOBAL)
import gsx  # noqa: F401
from gsx import ops


def timeit(fn, n=50, reps=5):
    for _ in range(5):
        fn()
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / n)
    return sorted(out)[len(out) // 2]


H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1080, 1920)
r = torch.rand(1, H, W, 3, device="cuda")
gt = torch.rand(1, 3, H, W, device="cuda")
l3, ws = ops.photometric_loss_fwd(r, gt, 0.2)
print("%s  loss fwd %.4f ms  bwd %.4f ms  loss %.6f" % (os.environ.get("TAG", ""), timeit(lambda: ops.photometric_loss_fwd(r, gt, 0.2)),
                                                       timeit(lambda: ops.photometric_loss_bwd(r, gt, ws, 0.2, None, 1.0)), float(l3[0])))
