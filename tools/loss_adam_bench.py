"""Times the next-tier kernels (photometric loss fwd/bwd, fused Adam over the S-1M parameter set) with HIP events."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gsx  # noqa: F401
from gsx import ops, loss

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

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1080, 1920)
r = torch.rand(1, H, W, 3, device="cuda")
gt = torch.rand(1, 3, H, W, device="cuda")
l3, ws = ops.photometric_loss_fwd(r, gt, 0.2)
print("loss fwd  %.3f ms" % timeit(lambda: ops.photometric_loss_fwd(r, gt, 0.2)))
print("loss bwd  %.3f ms" % timeit(lambda: ops.photometric_loss_bwd(r, gt, ws, 0.2, None, 1.0)))
img = r[0].permute(2, 0, 1).contiguous().unsqueeze(0)
m = ops.fusedssim(1e-4, 9e-4, img, gt, True)
print("fusedssim fwd %.3f ms" % timeit(lambda: ops.fusedssim(1e-4, 9e-4, img, gt, True)))
print("fusedssim bwd %.3f ms" % timeit(lambda: ops.fusedssim_backward(1e-4, 9e-4, img, gt, m[0], m[1], m[2], m[3])))
def ref_loss():
    R = r.detach().requires_grad_(True)
    v = loss.photometric_loss_reference(R.clamp(0, 1).permute(0, 3, 1, 2), gt, 0.2)
    v.backward()
def fused_loss():
    R = r.detach().requires_grad_(True)
    loss.photometric_loss(R, gt, 0.2).backward()
print("op-by-op loss fwd+bwd (autograd) %.3f ms" % timeit(ref_loss))
print("fused loss fwd+bwd (autograd)    %.3f ms" % timeit(fused_loss))
N = 1_000_000
shapes = {"means": (N, 3), "sh": (N, 16, 3), "scaling": (N, 3), "rotation": (N, 4), "opacity": (N, 1)}
P = {k: torch.randn(s, device="cuda") for k, s in shapes.items()}
G = {k: torch.randn(s, device="cuda") for k, s in shapes.items()}
M = {k: torch.zeros(s, device="cuda") for k, s in shapes.items()}
V = {k: torch.zeros(s, device="cuda") for k, s in shapes.items()}
M["sh0"], V["sh0"], M["shN"], V["shN"] = (torch.zeros(N, 1, 3, device="cuda"), torch.zeros(N, 1, 3, device="cuda"),
                                          torch.zeros(N, 15, 3, device="cuda"), torch.zeros(N, 15, 3, device="cuda"))
def adam_all():
    for k in ("means", "scaling", "rotation", "opacity"):
        ops.adam_step(P[k], M[k], V[k], G[k], 1e-3, 0.9, 0.999, 1e-8, 10.0, 31.6)
    ops.adam_step(P["sh"][:, :1], M["sh0"], V["sh0"], G["sh"][:, :1], 1e-3, 0.9, 0.999, 1e-8, 10.0, 31.6)
    ops.adam_step(P["sh"][:, 1:], M["shN"], V["shN"], G["sh"][:, 1:], 1e-3, 0.9, 0.999, 1e-8, 10.0, 31.6)
t = timeit(adam_all)
byts = 59 * N * 4 * 7  # read p,m,v,g; write p,m,v
print("adam 6 groups, 1M gaussians: %.3f ms  (%.0f GB/s of %d MB)" % (t, byts / t / 1e6, byts // 2**20))
Msh, Vsh = torch.zeros(N, 16, 3, device="cuda"), torch.zeros(N, 16, 3, device="cuda")
def adam_all_split():
    for k in ("means", "scaling", "rotation", "opacity"):
        ops.adam_step(P[k], M[k], V[k], G[k], 1e-3, 0.9, 0.999, 1e-8, 10.0, 31.6)
    ops.adam_step_split(P["sh"], Msh, Vsh, G["sh"], 3, 1e-3, 5e-5, True, True, 0.9, 0.999, 1e-8, 10.0, 31.6)
t = timeit(adam_all_split)
print("adam 5 launches (split SH), 1M gaussians: %.3f ms  (%.0f GB/s)" % (t, byts / t / 1e6))
for k in ("means", "rotation", "opacity"):
    t = timeit(lambda: ops.adam_step(P[k], M[k], V[k], G[k], 1e-3, 0.9, 0.999, 1e-8, 10.0, 31.6))
    print("  %-9s %.4f ms (%.0f GB/s)" % (k, t, P[k].numel() * 28 / t / 1e6))
t = timeit(lambda: ops.adam_step_split(P["sh"], Msh, Vsh, G["sh"], 3, 1e-3, 5e-5, True, True, 0.9, 0.999, 1e-8, 10.0, 31.6))
print("  %-9s %.4f ms (%.0f GB/s)" % ("sh split", t, P["sh"].numel() * 28 / t / 1e6))
