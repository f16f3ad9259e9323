"""Host time of one S-1M training iteration: wall time of the enqueueing loop minus the time the host spends waiting for the GPU
(inside IsectLists.confirm(), the one place it waits).  If that is well below the GPU's ~1.25 ms per iteration the loop is GPU-bound
with room to spare on a slower / shared host.  python tools/host_time.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gsx  # noqa: F401
from gsx import distributed as gdist, layout, loss as gloss, ops, optim, rasterizer, scenes
import gc
dev = "cuda:0"
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
sc = scenes.scene_1m()
order = layout.morton_order(sc["means"])
for k in ("means", "quats", "scales", "opacities", "sh"):
    sc[k] = sc[k][order].contiguous()
model = scenes.to_splat_data(sc, dev)
for p in model.params():
    p.requires_grad_(True)
cam = rasterizer.Camera(viewmat=sc["viewmat"].to(dev), K=sc["K"].to(dev), width=sc["width"], height=sc["height"])
bucket = gdist.GradBucket(model.params())
sinks = bucket.sinks()
opt = optim.FusedAdam.for_splat_data(model)
bg = sc["background"].to(dev)
target = torch.rand(3, sc["height"], sc["width"], device=dev)
wait = [0.0]
orig_confirm = ops._C.IsectLists.confirm
def timed_confirm(self):
    t = time.perf_counter()
    r = orig_confirm(self)
    wait[0] += time.perf_counter() - t
    return r
ops._C.IsectLists.confirm = timed_confirm
def step(i):
    sinks["_sh_adam"] = opt.begin_fused_sh_step(1001 + i)
    out = rasterizer.rasterize_fused(cam, model, bg, grad_sinks=sinks, guarded=True)
    loss = gloss.photometric_loss(out.render_hwc, target, 0.2)
    gloss.backward(loss)
    opt.step(1001 + i, skip_sh=True)
for i in range(20):
    step(i)
torch.cuda.synchronize()
for label, prep in (("gc enabled", lambda: gc.enable()), ("gc disabled", lambda: gc.disable())):
    prep()
    wait[0] = 0.0
    per = []
    t0 = time.perf_counter()
    for i in range(steps):
        t = time.perf_counter()
        step(100 + i)
        per.append(time.perf_counter() - t)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    per.sort()
    print("%-12s wall %.4f ms/step, host waiting %.4f ms/step -> host busy %.4f ms/step; slowest enqueue %.3f ms, p99 %.3f ms"
          % (label, wall / steps * 1e3, wait[0] / steps * 1e3, (wall - wait[0]) / steps * 1e3, per[-1] * 1e3, per[int(0.99 * steps)] * 1e3))
