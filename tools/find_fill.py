"""Debug tool: which Python line launches the per-iteration at::native::FillFunctor kernel of the S-1M training step?
torch.profiler with stacks over three iterations of bench.py's step; prints every fill-like kernel's launching op and Python stack.
GPU box: python tools/find_fill.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import gsx  # noqa: E402,F401
from gsx import distributed as gdist  # noqa: E402
from gsx import loss as gloss  # noqa: E402
from gsx import optim, rasterizer, scenes  # noqa: E402

dev = "cuda:0"
scene = scenes.scene_1m()
model = scenes.to_splat_data(scene, dev)
for p in model.params():
    p.requires_grad_(True)
names = ["means", "sh", "scaling_raw", "rotation_raw", "opacity_raw"]
bucket = gdist.GradBucket([getattr(model, n) for n in names])
sinks = bucket.sinks(tuple(names))
opt = optim.FusedAdam.for_splat_data(model)
cam = rasterizer.Camera(viewmat=scene["viewmat"].to(dev), K=scene["K"].to(dev), width=scene["width"], height=scene["height"])
bg = scene["background"].to(dev)
target = torch.rand(3, scene["height"], scene["width"], device=dev)


def step(i):
    sinks["_sh_adam"] = opt.begin_fused_sh_step(1001 + i)
    out = rasterizer.rasterize_fused(cam, model, bg, grad_sinks=sinks, guarded=True)
    loss = gloss.photometric_loss(out.render_hwc, target, 0.2)
    gloss.backward(loss)
    opt.step(1001 + i, skip_sh=sinks["_sh_adam"] is not None)


for i in range(5):
    step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for i in range(5, 8):
        step(i)
    torch.cuda.synchronize()
seen = 0
for ev in prof.events():
    n = ev.name
    if ("fill" in n.lower() or "zero" in n.lower() or "ones" in n.lower()) and ev.device_type == torch.autograd.DeviceType.CPU:
        seen += 1
        print("OP", n, "shapes", ev.input_shapes, "cuda_time_us", getattr(ev, "device_time_total", None))
        for fr in (ev.stack or [])[:12]:
            print("     ", fr)
print("fill-like ops in 3 iterations:", seen)
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=70))
