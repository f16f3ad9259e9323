"""Stand-in for BASELINE configs[2] ("Mip-NeRF360 'garden', default_optimization_params.json training loop, 1xMI355X") on a box without
datasets: a synthetic scene of garden's shape — 185 cameras on three rings around a central cluster, 1296x840 (garden's `images_4`
size), focal 960 — trained with the `--gut` path's training step (rasterize_fused -> fused L1+SSIM loss -> backward -> MCMC strategy ->
fused Adam) and the hyper-parameters of the file configs[2] names (gsx/parameters.py preset "default": means_lr 1.6e-5, no regularisers,
start_refine 500, refine_every 100, max_cap 1M), strategy forced to MCMC (the only densification that gets a signal under --gut:
SURVEY §8f-3).  Ground-truth images are renders of a hidden 400 k-Gaussian scene; the trainee starts from 200 k of its points
(positions jittered, as an SfM cloud would be) through init_model_from_pointcloud and grows to the 1 M cap.

    python examples/train_garden_standin.py [iterations=4000] [--json out.json] [--profile-ops]

Prints one JSON line: iterations/s (whole loop, incl. densification), PSNR before / after over 24 held-in cameras, Gaussian count.
"""
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import gsx  # noqa: E402,F401
from gsx import io_colmap, metrics, parameters, rasterizer, scenes, trainer  # noqa: E402

W, H, FOCAL = 1296, 840, 960.0
N_CAMS, N_GT, N_INIT = 185, 400_000, 200_000


def hidden_scene(dev, seed=7):
    """Central cluster (a 'table' of radius 1.2) + a ground disc + a far ring: anisotropic Gaussians, SH degree 3."""
    g = torch.Generator().manual_seed(seed)
    n_obj, n_ground = N_GT // 2, N_GT // 4
    n_far = N_GT - n_obj - n_ground
    obj = torch.randn(n_obj, 3, generator=g) * torch.tensor([0.7, 0.45, 0.7])
    r = torch.sqrt(torch.rand(n_ground, generator=g)) * 4.0
    a = torch.rand(n_ground, generator=g) * 2 * math.pi
    ground = torch.stack([r * torch.cos(a), 0.9 + 0.03 * torch.randn(n_ground, generator=g), r * torch.sin(a)], 1)   # +y is down (OpenCV)
    a = torch.rand(n_far, generator=g) * 2 * math.pi
    far = torch.stack([9.0 * torch.cos(a), -1.5 + 2.5 * torch.rand(n_far, generator=g), 9.0 * torch.sin(a)], 1)
    means = torch.cat([obj, ground, far], 0)
    lo, hi = math.log(0.006), math.log(0.06)
    scales = torch.exp(torch.rand(N_GT, 3, generator=g) * (hi - lo) + lo)
    scales[n_obj + n_ground:] *= 4.0
    quats = torch.nn.functional.normalize(torch.randn(N_GT, 4, generator=g), dim=-1)
    opac = torch.rand(N_GT, generator=g) * 0.7 + 0.25
    sh = torch.zeros(N_GT, 16, 3)
    sh[:, 0] = (torch.rand(N_GT, 3, generator=g) - 0.5) * 2.5          # DC: colours over most of [0, 1]
    sh[:, 1:] = (torch.rand(N_GT, 15, 3, generator=g) - 0.5) * 0.25     # view-dependent part
    sc = dict(means=means, quats=quats, scales=scales, opacities=opac, sh=sh, sh_degree=3)
    return scenes.to_splat_data(sc, dev)


def ring_cameras(dev):
    cams = []
    K = scenes.intrinsics(FOCAL, FOCAL, W / 2.0, H / 2.0).to(dev)
    for i in range(N_CAMS):
        ring = i % 3
        a = 2 * math.pi * (i // 3) / math.ceil(N_CAMS / 3) + 0.11 * ring
        radius, height = (4.2, 4.8, 5.4)[ring], (-0.6, -1.6, -2.6)[ring]
        eye = (radius * math.sin(a), height, -radius * math.cos(a))
        vm = scenes.look_at_viewmat(eye, (0.0, 0.1, 0.0))
        cams.append(rasterizer.Camera(viewmat=vm.to(dev), K=K, width=W, height=H))
    return cams


def setup(dev="cuda:0"):
    """Hidden scene -> ground-truth images -> SfM-like initial model -> Trainer with the 'default' preset (MCMC)."""
    bg = torch.zeros(3, device=dev)
    gt = hidden_scene(dev)
    cams = ring_cameras(dev)
    with torch.no_grad():
        images = [rasterizer.rasterize_fused(c, gt, bg).image.clone() for c in cams]
    # the SfM-like initial cloud: a subset of the hidden scene's centres, jittered; colours from the DC term
    g = torch.Generator().manual_seed(11)
    pick = torch.randperm(N_GT, generator=g)[:N_INIT]
    pts = (gt.means[pick.to(dev)].cpu() + 0.01 * torch.randn(N_INIT, 3, generator=g)).numpy()
    rgb = ((gt.sh[pick.to(dev), 0].cpu() * 0.28209479177387814 + 0.5).clamp(0, 1) * 255).numpy().astype(np.uint8)
    params = parameters.OptimizationParameters.preset("default")      # the file configs[2] names
    params.strategy = "mcmc"
    model, scene_scale = io_colmap.init_model_from_pointcloud(pts, rgb, (0.0, 0.0, 0.0), sh_degree=3, init_scaling=params.init_scaling,
                                                             init_opacity=params.init_opacity, device=dev)
    tr = trainer.Trainer(model, cams, images, params, bg, scene_scale=scene_scale, seed=0)
    return tr, model, cams, images, bg, params


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    iters = int(args[0]) if args else 4000
    out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    tr, model, cams, images, bg, params = setup()
    eval_idx = list(range(0, N_CAMS, 8))
    ev = lambda: metrics.evaluate(model, [cams[i] for i in eval_idx], [images[i] for i in eval_idx], bg)  # noqa: E731
    before = ev()
    import gc
    gc.collect()
    gc.freeze()   # as Trainer.train(): the dataset and modules leave the collector's young generations — its collections inside the loop stay short
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks = {}
    for it in range(1, iters + 1):
        tr.train_step(it)
        if it in (iters // 4, iters // 2, 3 * iters // 4):
            torch.cuda.synchronize()
            marks[it] = (time.perf_counter() - t0, model.means.shape[0])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    after = ev()
    # steady state at the final model size: the last quarter of the run
    t34, _ = marks[3 * iters // 4]
    res = {"workload": "garden stand-in (BASELINE configs[2]): %d ring cameras @%dx%d, hidden scene of %d Gaussians, init %d, MCMC to max_cap %d, "
                       "parameter preset 'default' (default_optimization_params.json)" % (N_CAMS, W, H, N_GT, N_INIT, params.max_cap),
           "iterations": iters, "seconds": round(dt, 2), "iters_per_s": round(iters / dt, 1),
           "iters_per_s_last_quarter": round((iters - 3 * iters // 4) / (dt - t34), 1),
           "gaussians_start": N_INIT, "gaussians_end": int(model.means.shape[0]),
           "gaussians_at": {str(k): v[1] for k, v in marks.items()},
           "psnr_before": round(before["psnr"], 2), "psnr_after": round(after["psnr"], 2),
           "ssim_before": round(before["ssim"], 4), "ssim_after": round(after["ssim"], 4),
           "active_sh_degree": model.active_sh_degree, "means_lr_end": tr.strategy.optimizer.groups[0]["lr"],
           "intersect_protocol": "guarded lists" if tr.guarded else "exact", "iterations_repeated_for_list_capacity": int(tr.capacity_misses),
           "non_finite_gaussians_relocated": int(getattr(tr.strategy, "nonfinite_relocated", 0))}
    if "--profile-ops" in sys.argv:   # per-operator HIP-event times of 48 more iterations at the final model size (bench.OpTimer)
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        from gsx import ops
        timer = bench.OpTimer(ops)
        n_prof = 48 if (iters + 48) < params.iterations else 0
        timer.enabled = True
        t1 = time.perf_counter()
        for it in range(iters + 1, iters + 1 + n_prof):
            tr.train_step(it)
        torch.cuda.synchronize()
        timer.enabled = False
        res["ops_ms"] = {k: round(v, 4) for k, v in sorted(timer.mean_ms().items(), key=lambda kv: -kv[1])}
        res["ops_ms_iteration_wall"] = round((time.perf_counter() - t1) / max(1, n_prof) * 1e3, 4)
    line = json.dumps(res)
    print(line)
    if out_json:
        with open(out_json, "w") as f:
            f.write(line + "\n")


if __name__ == "__main__":
    main()
