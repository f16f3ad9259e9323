"""Long-run check of examples/train_synthetic.py's setting: non-finite parameters, PLY round trip field by field.   python tools/soak_check.py [iterations]"""
import math, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gsx  # noqa: F401
from gsx import io_ply, metrics, rasterizer, scenes, strategy, trainer

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
dev = "cuda:0"
sc = scenes.scene_small(seed=5, N=20000)
gt = scenes.to_splat_data(sc, dev)
W, H = sc["width"], sc["height"]
cams = []
for k in range(8):
    vm = sc["viewmat"].clone()
    vm[0, 3] += 0.2 * math.cos(k * math.pi / 4); vm[1, 3] += 0.2 * math.sin(k * math.pi / 4)
    cams.append(rasterizer.Camera(viewmat=vm.to(dev), K=sc["K"].to(dev), width=W, height=H))
bg = sc["background"].to(dev)
with torch.no_grad():
    images = [rasterizer.rasterize_fused(c, gt, bg).image.clone() for c in cams]
g = torch.Generator().manual_seed(1)
model = scenes.to_splat_data(dict(sc), dev)
model.sh = (gt.sh + 0.3 * torch.randn(gt.sh.shape, generator=g).to(dev)).contiguous()
model.means = (gt.means + 0.01 * torch.randn(gt.means.shape, generator=g).to(dev)).contiguous()
params = strategy.OptimizationParameters(iterations=iters, start_refine=iters // 8, refine_every=max(1, iters // 8), stop_refine=iters, max_cap=22000)
tr = trainer.Trainer(model, cams, images, params, bg, seed=0)
mode = sys.argv[2] if len(sys.argv) > 2 else "nosync"
flags = []
for it in range(1, iters + 1):
    tr.train_step(it)
    if it % 250 == 0:
        f = torch.stack([(~torch.isfinite(getattr(model, n))).sum() for n in ("means", "sh", "scaling_raw", "rotation_raw", "opacity_raw")])
        flags.append((it, model.means.shape[0], f))       # device tensor: read at the end (no synchronisation inside the run)
        if mode == "sync":
            torch.cuda.synchronize()
torch.cuda.synchronize()
prev = 0
for it, n, f in flags:
    tot = int(f.sum())
    if tot != prev:
        print("iteration %6d: N %d, non-finite elements (means, sh, scaling, rotation, opacity) = %s" % (it, n, f.tolist()), flush=True)
        prev = tot
print("mode %s: %d iterations, N %d, non-finite at the end: %d" % (mode, iters, model.means.shape[0], prev), flush=True)
with tempfile.TemporaryDirectory() as d:
    path = io_ply.save_ply(model, d, iteration=iters)
    back = io_ply.load_ply(path, dev)
    for n in ("means", "sh", "scaling_raw", "rotation_raw", "opacity_raw"):
        a, b = getattr(model, n).detach(), getattr(back, n)
        if n == "sh":
            b = b[:, :a.shape[1]]
        b = b.reshape(a.shape)
        d = (a - b).abs()
        print(n, tuple(a.shape), "equal" if torch.equal(a, b) else "DIFFERENT: %d elements, max |d| %.3g (at value %.6g vs %.6g), non-finite %d / %d" % (
            int((a != b).sum()), float(d.nan_to_num(0).max()), float(a.flatten()[d.nan_to_num(0).argmax()]), float(b.flatten()[d.nan_to_num(0).argmax()]),
            int((~torch.isfinite(a)).sum()), int((~torch.isfinite(b)).sum())))
