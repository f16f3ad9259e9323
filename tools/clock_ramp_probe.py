import sys, time, os
sys.path.insert(0, '/root/repo')
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import gsx
from gsx import loss as gloss, optim, rasterizer, scenes, layout, distributed as gdist
dev = torch.device('cuda', 0)
scene = scenes.scene_1m()
order = layout.morton_order(scene["means"])
for k in ("means", "quats", "scales", "opacities", "sh"):
    scene[k] = scene[k][order].contiguous()
model = scenes.to_splat_data(scene, dev)
for p in model.params(): p.requires_grad_(True)
opt = optim.FusedAdam.for_splat_data(model)
names = ["sh", "means", "scaling_raw", "rotation_raw", "opacity_raw"]
bucket = gdist.GradBucket([getattr(model, n) for n in names]); sinks = bucket.sinks(tuple(names))
cam = rasterizer.Camera(viewmat=scene["viewmat"].to(dev), K=scene["K"].to(dev), width=1920, height=1080)
bg = scene["background"].to(dev)
target = torch.rand(3, 1080, 1920).to(dev)
snap = [p.detach().clone() for p in model.params()]
def step(i):
    sinks["_sh_adam"] = opt.begin_fused_sh_step(1001 + i)
    out = rasterizer.rasterize_fused(cam, model, bg, grad_sinks=sinks, guarded=True)
    l = gloss.photometric_loss(out.render_hwc, target, 0.2)
    gloss.backward(l)
    opt.step(1001 + i, skip_sh=True)
for i in range(10): step(i)
torch.cuda.synchronize()
for gap_ms in (0, 2, 20, 200):
    with torch.no_grad():
        for p, q in zip(model.params(), snap): p.copy_(q)
    for i in range(30): step(i)
    torch.cuda.synchronize()
    time.sleep(gap_ms * 1e-3)
    ts = []
    t0 = time.perf_counter()
    for blk in range(12):
        for i in range(10): step(i)
        torch.cuda.synchronize()
        ts.append(time.perf_counter())
    d = [(ts[0] - t0) / 10 * 1e3] + [(ts[k] - ts[k - 1]) / 10 * 1e3 for k in range(1, len(ts))]
    print("idle gap %3d ms before: ms/step per block of 10:" % gap_ms, " ".join("%.3f" % x for x in d))
# no syncs inside: 20 steps right after 30 warm steps + one sync
for rep in range(3):
    with torch.no_grad():
        for p, q in zip(model.params(), snap): p.copy_(q)
    for i in range(30): step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20): step(i)
    torch.cuda.synchronize()
    print("20 steps right behind 30 warm steps (fixed camera): %.4f ms/step" % ((time.perf_counter() - t0) / 20 * 1e3))
