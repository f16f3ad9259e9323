"""Which host op launches a fill kernel inside the training step?  python tools/find_fill.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import gsx  # noqa: F401
from gsx import distributed as gdist, loss as gloss, optim, rasterizer, scenes
dev = "cuda:0"
sc = scenes.scene_small(seed=1, N=20000)
g = torch.Generator().manual_seed(5)
sc["sh"] = (torch.rand(20000, 16, 3, generator=g) - 0.5) * 0.6
sc["sh_degree"] = 3
model = scenes.to_splat_data(sc, dev)
for p in model.params():
    p.requires_grad_(True)
cam = rasterizer.Camera(viewmat=sc["viewmat"].to(dev), K=sc["K"].to(dev), width=sc["width"], height=sc["height"])
bucket = gdist.GradBucket(model.params())
sinks = bucket.sinks()
opt = optim.FusedAdam.for_splat_data(model)
bg = sc["background"].to(dev)
target = torch.rand(3, sc["height"], sc["width"], device=dev)
def step(i):
    sinks["_sh_adam"] = opt.begin_fused_sh_step(1001 + i)
    out = rasterizer.rasterize_fused(cam, model, bg, grad_sinks=sinks, guarded=True)
    loss = gloss.photometric_loss(out.render_hwc, target, 0.2)
    gloss.backward(loss)
    opt.step(1001 + i, skip_sh=True)
for i in range(5):
    step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(6)
    torch.cuda.synchronize()
names = [e.name for e in prof.events() if e.device_type is not None and str(e.device_type).endswith("CUDA")]
print("device kernels of the step:", [n[:40] for n in names])
for e in prof.events():
    if "fill" in e.name.lower() or "zero" in e.name.lower() or "ones" in e.name.lower() or "elementwise" in e.name.lower():
        par = e.cpu_parent
        chain = []
        while par is not None and len(chain) < 6:
            chain.append(par.name)
            par = par.cpu_parent
        print(e.name, "<-", " <- ".join(chain), "| stack:", [s for s in (e.stack or [])[:6]])
