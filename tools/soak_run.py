"""examples/train_synthetic.py's run with the trainer's switches from the environment (SOAK_GUARDED / SOAK_FUSED_SH / SOAK_FUSED_REG = 0 | 1, SOAK_REFINE = 0: no refine
events, SOAK_CAMERA = pinhole | fisheye | distorted, SOAK_INSIDE = 1: cameras inside the cloud), ending in one line: ok | non-finite | (a crash prints nothing).    python tools/soak_run.py [iterations]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gsx  # noqa: F401
from gsx import rasterizer, scenes, strategy, trainer

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
env = lambda k: os.environ.get(k, "1") != "0"  # noqa: E731
dev = "cuda:0"
sc = scenes.scene_small(seed=5, N=20000)
gt = scenes.to_splat_data(sc, dev)
W, H = sc["width"], sc["height"]
cams = []
for k in range(8):
    vm = sc["viewmat"].clone()
    vm[0, 3] += 0.2 * math.cos(k * math.pi / 4); vm[1, 3] += 0.2 * math.sin(k * math.pi / 4)
    if os.environ.get("SOAK_INSIDE"):   # the cameras INSIDE the cloud (its depths are 2 .. 3): Gaussians at, around and behind the camera plane, screen-filling footprints
        vm[2, 3] -= 2.3 + 0.05 * k
    # SOAK_CAMERA=fisheye | distorted: the same poses through an equidistant fisheye / a pinhole with radial + tangential distortion (the fast path's chart flags and
    # generic tiles, resp. the distorted projection, under 30 000 iterations of MCMC)
    kind = os.environ.get("SOAK_CAMERA", "pinhole")
    extra = {}
    if kind == "fisheye":
        from gsx import ops
        extra = dict(camera_model=ops.CameraModelType.FISHEYE, radial=torch.tensor([0.01, -0.002, 0.0, 0.0], device=dev))
    elif kind == "distorted":
        extra = dict(radial=torch.tensor([0.05, -0.01, 0.002, 0.0], device=dev), tangential=torch.tensor([0.002, -0.001], device=dev))
    cams.append(rasterizer.Camera(viewmat=vm.to(dev), K=sc["K"].to(dev), width=W, height=H, **extra))
bg = sc["background"].to(dev)
with torch.no_grad():
    images = [rasterizer.rasterize_fused(c, gt, bg).image.clone() for c in cams]
g = torch.Generator().manual_seed(1)
model = scenes.to_splat_data(dict(sc), dev)
model.sh = (gt.sh + 0.3 * torch.randn(gt.sh.shape, generator=g).to(dev)).contiguous()
model.means = (gt.means + 0.01 * torch.randn(gt.means.shape, generator=g).to(dev)).contiguous()
refine = env("SOAK_REFINE")
params = strategy.OptimizationParameters(iterations=iters, start_refine=iters // 8 if refine else 10 * iters, refine_every=max(1, iters // 8), stop_refine=iters, max_cap=22000)
tr = trainer.Trainer(model, cams, images, params, bg, seed=0, fused_sh_adam=env("SOAK_FUSED_SH"), guarded_lists=env("SOAK_GUARDED"), fused_regularisers=env("SOAK_FUSED_REG"))
if os.environ.get("SOAK_DEBUG"):   # where do the first non-finite values come from? (synchronises: a debugging run)
    st = tr.strategy
    orig_rel = st._relocated
    names = ("means", "sh", "scaling_raw", "rotation_raw", "opacity_raw")
    state = {"it": 0, "reported": False}

    def nonfinite():
        return {n: int((~torch.isfinite(getattr(model, n))).sum()) for n in names}

    def rel(opacities, sampled, ratios):
        new_op, new_sc = orig_rel(opacities, sampled, ratios)
        bad = (~torch.isfinite(new_sc)).any(-1) | (new_sc <= 0).any(-1) | ~torch.isfinite(new_op)
        if bool(bad.any()) and not state["reported"]:
            k = bad.nonzero().flatten()[:6]
            print("iteration %d: relocation returned %d bad rows; e.g. opacity in %s ratios %s -> new opacity %s new scales %s (old scales %s)" % (
                state["it"], int(bad.sum()), opacities.index_select(0, sampled)[k].tolist(), ratios[k].tolist(), new_op[k].tolist(), new_sc[k].tolist(),
                model.get_scaling().index_select(0, sampled)[k].tolist()), flush=True)
        return new_op, new_sc
    st._relocated = rel
    orig_step = st.step
    last_out = {}
    orig_pb = st.post_backward

    def pb(it, out=None):
        last_out["out"] = out
        return orig_pb(it, out)
    st.post_backward = pb

    def step_hook(it, optimizer_step=None):
        flat = tr.bucket.flat
        if not state["reported"] and not bool(torch.isfinite(flat).all()):
            state["reported"] = True
            g = {n: getattr(model, n).grad for n in names}
            nf = {n: int((~torch.isfinite(g[n])).sum()) for n in names if g[n] is not None}
            bad = (~torch.isfinite(g["scaling_raw"])).any(-1) | (~torch.isfinite(g["means"])).any(-1)
            idx = bad.nonzero().flatten()
            out = last_out.get("out")
            print("iteration %d: first non-finite GRADIENTS %s; %d Gaussians" % (it, nf, idx.numel()), flush=True)
            k = idx[:8]
            sc = model.get_scaling()[k]; op = model.get_opacity()[k].flatten()
            print("   Gaussians %s\n   opacity %s\n   scales %s\n   |rotation_raw| %s\n   means %s" % (k.tolist(), op.tolist(), sc.tolist(), model.rotation_raw[k].norm(dim=-1).tolist(), model.means[k].tolist()), flush=True)
            if out is not None:
                m2d = out.means2d.reshape(-1, 2)[idx]
                rad = out.aux["radii_full"].reshape(-1, 2)[idx]
                print("   means2d of the bad Gaussians: x %.1f .. %.1f, y %.1f .. %.1f; radii max %s; depths %.3f .. %.3f" % (float(m2d[:, 0].min()), float(m2d[:, 0].max()), float(m2d[:, 1].min()),
                      float(m2d[:, 1].max()), rad.max(0).values.tolist(), float(out.depths.reshape(-1)[idx].min()), float(out.depths.reshape(-1)[idx].max())), flush=True)
                print("   render finite: %s, alpha range %.4f .. %.4f" % (bool(torch.isfinite(out.render_hwc).all()), float(out.alpha.min()), float(out.alpha.max())), flush=True)
                allsc = model.get_scaling(); allop = model.get_opacity().flatten()
                print("   model: scale range %.3g .. %.3g, scale ratio max %.3g, opacity range %.3g .. %.3g, |rot| range %.3g .. %.3g" % (float(allsc.min()), float(allsc.max()),
                      float((allsc.max(-1).values / allsc.min(-1).values).max()), float(allop.min()), float(allop.max()), float(model.rotation_raw.norm(dim=-1).min()), float(model.rotation_raw.norm(dim=-1).max())), flush=True)
        return orig_step(it, optimizer_step)
    st.step = step_hook
    for it in range(1, iters + 1):
        state["it"] = it
        tr.train_step(it)
        if (st.is_refining(it) or it % 250 == 0) and not state["reported"]:
            nf = nonfinite()
            if sum(nf.values()):
                state["reported"] = True
                print("iteration %d (%s): first non-finite parameters %s" % (it, "refine" if st.is_refining(it) else "plain", nf), flush=True)
                idx = (~torch.isfinite(model.scaling_raw)).any(-1).nonzero().flatten()[:4]
                print("   e.g. Gaussians %s: opacity_raw %s scaling_raw %s" % (idx.tolist(), model.opacity_raw[idx].flatten().tolist(), model.scaling_raw[idx].tolist()), flush=True)
                break
else:
    tr.train(iters, log_every=0)
torch.cuda.synchronize()
bad = sum(int((~torch.isfinite(getattr(model, n))).sum()) for n in ("means", "sh", "scaling_raw", "rotation_raw", "opacity_raw"))
print("soak: %s (N %d, %d non-finite elements, capacity misses %d)" % ("ok" if bad == 0 else "non-finite", model.means.shape[0], bad, tr.capacity_misses), flush=True)
