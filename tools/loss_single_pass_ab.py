"""Round 6: the single-pass training loss (gsx_photometric_loss_single_pass) against the forward + backward pair, same process: values and time.
   python tools/loss_single_pass_ab.py [H W]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
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
g = torch.Generator(device="cuda").manual_seed(1)
r = torch.rand(1, H, W, 3, device="cuda", generator=g) * 1.2 - 0.1      # some pixels outside [0, 1]: the clamp mask
gt = torch.rand(1, 3, H, W, device="cuda", generator=g)
l3, ws = ops.photometric_loss_fwd(r, gt, 0.2)
v = ops.photometric_loss_bwd(r, gt, ws, 0.2, None, 1.0)
l3s, vs = ops.photometric_loss_single_pass(r, gt, 0.2, 1.0)
torch.cuda.synchronize()
print("loss pair %s single %s   max |v_pair - v_single| = %.3e (max |v| %.3e)  equal bits: %s" % (l3.tolist(), l3s.tolist(), float((v - vs).abs().max()), float(v.abs().max()), bool(torch.equal(v, vs))))


def pair():
    a, w_ = ops.photometric_loss_fwd(r, gt, 0.2)
    return ops.photometric_loss_bwd(r, gt, w_, 0.2, None, 1.0)


for rep in range(3):
    print("pair (fwd + finalize + bwd) %.4f ms   single pass (+ finalize) %.4f ms" % (timeit(pair), timeit(lambda: ops.photometric_loss_single_pass(r, gt, 0.2, 1.0))))
