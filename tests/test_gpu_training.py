"""End-to-end check of the training step (render -> fused L1/SSIM loss -> backward -> regularisers -> MCMC strategy -> fused
Adam): a perturbed copy of a small scene is trained against renders of the original from four cameras and must converge."""
import math

import pytest
import torch


def _scene(dev, N=3000, K=4):
    import gsx  # noqa: F401
    from gsx import rasterizer, scenes
    sc = scenes.scene_small(seed=5, N=N)
    g = torch.Generator().manual_seed(3)
    sc["sh"] = torch.cat([sc["sh"][:, :1], 0.05 * torch.randn(N, K - 1, 3, generator=g)], 1) if sc["sh"].shape[1] < K else sc["sh"]
    sc["sh_degree"] = int(math.isqrt(K)) - 1
    model = scenes.to_splat_data(sc, dev)
    W, H = sc["width"], sc["height"]
    cams = []
    for k in range(4):
        vm = sc["viewmat"].clone()
        vm[0, 3] += 0.15 * math.cos(k * math.pi / 2)
        vm[1, 3] += 0.15 * math.sin(k * math.pi / 2)
        cams.append(rasterizer.Camera(viewmat=vm.to(dev), K=sc["K"].to(dev), width=W, height=H))
    return sc, model, cams


@pytest.mark.gpu
def test_training_converges_and_mcmc_bookkeeping():
    import gsx  # noqa: F401
    from gsx import rasterizer, scenes, strategy, trainer
    dev = "cuda:0"
    sc, gt_model, cams = _scene(dev)
    bg = sc["background"].to(dev)
    with torch.no_grad():
        images = [rasterizer.rasterize_fused(c, gt_model, bg).image.clone() for c in cams]
    # the trainee: the same scene with perturbed colours, positions and opacities
    g = torch.Generator().manual_seed(9)
    sc2 = dict(sc)
    model = scenes.to_splat_data(sc2, dev)
    model.sh = (gt_model.sh + 0.3 * torch.randn(gt_model.sh.shape, generator=g).to(dev)).contiguous()
    model.means = (gt_model.means + 0.01 * torch.randn(gt_model.means.shape, generator=g).to(dev)).contiguous()
    model.opacity_raw = (gt_model.opacity_raw - 0.5).contiguous()
    model.scaling_raw, model.rotation_raw = gt_model.scaling_raw.clone(), gt_model.rotation_raw.clone()
    model.active_sh_degree = gt_model.active_sh_degree
    params = strategy.OptimizationParameters(iterations=300, start_refine=50, refine_every=50, stop_refine=250, max_cap=3300,
                                             sh_degree_interval=1000)
    tr = trainer.Trainer(model, cams, images, params, bg, seed=1)
    tr.strategy.NOISE_LR = 0.0            # noise off for the convergence check (it is exercised separately below)
    first = float(sum(float(tr.train_step(it)) for it in range(1, 5)) / 4)
    n0 = model.means.shape[0]
    for it in range(5, 240):
        tr.train_step(it)
    last = float(sum(float(tr.train_step(it)) for it in range(240, 244)) / 4)
    assert last < 0.55 * first, (first, last)
    # growth by 5 % per refinement up to max_cap, moments extended with the model, step counts kept
    n1 = model.means.shape[0]
    assert n0 < n1 <= 3300
    opt = tr.strategy.optimizer
    assert opt.state["means"]["exp_avg"].shape[0] == n1 and opt.state["sh"]["exp_avg_sq"].shape[0] == n1
    # (the optimizer is not stepped on the iterations in which the model grew: the new tensors have no gradient, as upstream)
    n_growth = 0
    n = n0
    for it in range(1, 244):
        if tr.strategy.is_refining(it) and n < 3300:
            n_growth += 1
            n = min(3300, int(1.05 * n))
    assert n == n1 and n_growth >= 2
    assert opt.step_count("means") == 243 - n_growth and opt.step_count("shN") == 243 - n_growth
    assert all(p.grad is not None and p.grad.shape == p.shape for p in model.params())
    assert abs(opt.groups[0]["lr"] - 0.00016 * 0.01 ** (243 / 300)) < 1e-9


@pytest.mark.gpu
def test_mcmc_relocate_moves_dead_gaussians_onto_live_ones():
    import gsx  # noqa: F401
    from gsx import strategy
    dev = "cuda:0"
    sc, model, cams = _scene(dev, N=500)
    for p in model.params():
        p.requires_grad_(True)
    gen = torch.Generator(device=dev)
    gen.manual_seed(0)
    mc = strategy.MCMC(model, strategy.OptimizationParameters(), 1.0, gen)
    for p in model.params():
        p.grad = torch.ones_like(p)
    mc.optimizer.step(1)
    with torch.no_grad():
        model.opacity_raw[:40] = -10.0           # dead: sigmoid(-10) < min_opacity
    means_before = model.means.detach().clone()
    n_dead = mc.relocate_gs()
    assert n_dead == 40
    moved = model.means.detach()[:40]
    same = (moved[:, None, :] == means_before[None, 40:, :]).all(-1)      # [40, 460] exact matches
    assert bool(same.any(1).all())                      # every dead Gaussian now sits exactly on a live one
    assert float(torch.sigmoid(model.opacity_raw.detach()).min()) >= 0.005 - 1e-6
    # the moments of the sampled (source) Gaussians were reset
    src = same.float().argmax(dim=1) + 40
    assert float(mc.optimizer.state["means"]["exp_avg"][src].abs().max()) == 0.0
    # noise: displaces low-opacity Gaussians more than opaque ones, deterministic given the generator
    m0 = model.means.detach().clone()
    mc.inject_noise()
    assert float((model.means.detach() - m0).abs().max()) > 0.0


@pytest.mark.gpu
def test_mcmc_relocate_treats_non_finite_gaussians_as_dead(capfd):
    """Upstream has no such case (its multinomial asserts on a NaN opacity, mcmc.cpp:114-189): here a Gaussian holding a non-finite parameter is relocated
    like a dead one, its moments are cleared, and the event is reported on stderr; a finite model takes the unchanged path."""
    import gsx  # noqa: F401
    from gsx import strategy
    dev = "cuda:0"
    sc, model, cams = _scene(dev, N=500)
    for p in model.params():
        p.requires_grad_(True)
    gen = torch.Generator(device=dev)
    gen.manual_seed(0)
    mc = strategy.MCMC(model, strategy.OptimizationParameters(), 1.0, gen)
    for p in model.params():
        p.grad = torch.ones_like(p)
    mc.optimizer.step(1)
    assert mc.relocate_gs() == 0 and getattr(mc, "nonfinite_relocated", 0) == 0
    with torch.no_grad():
        model.means[3, 1] = float("nan")
        model.opacity_raw[7] = float("nan")
        model.scaling_raw[11, 0] = float("inf")
        model.sh[13, 2, 1] = float("nan")
        mc.optimizer.state["means"]["exp_avg"][3] = float("nan")
    assert mc.relocate_gs() == 4
    assert mc.nonfinite_relocated == 4
    assert "4 of 500 Gaussians hold non-finite parameters" in capfd.readouterr().err
    for p in model.params():
        assert bool(torch.isfinite(p.detach()).all())
    for st in (st for st in mc.optimizer.state.values() if isinstance(st, dict)):
        assert bool(torch.isfinite(st["exp_avg"]).all()) and bool(torch.isfinite(st["exp_avg_sq"]).all())
    assert float(mc.optimizer.state["means"]["exp_avg"][3].abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["1", "2"])
def test_check_finite_switches_name_the_first_non_finite_iteration(monkeypatch, mode):
    """GSX_CHECK_FINITE=1 (a synchronisation per iteration) / =2 (flag words read every 250 iterations): a clean run passes untouched, a NaN planted in a
    parameter is reported with the iteration it first showed up in."""
    import gsx  # noqa: F401
    from gsx import rasterizer, strategy, trainer
    monkeypatch.setenv("GSX_CHECK_FINITE", mode)
    dev = "cuda:0"
    sc, model, cams = _scene(dev, N=800)
    bg = sc["background"].to(dev)
    with torch.no_grad():
        images = [rasterizer.rasterize_fused(c, model, bg).image.clone() for c in cams]
    params = strategy.OptimizationParameters(iterations=600, start_refine=10_000, refine_every=100, stop_refine=20_000, max_cap=1000, sh_degree_interval=1000)
    tr = trainer.Trainer(model, cams, images, params, bg, seed=1)
    assert tr.check_finite == (mode == "1") and tr.check_finite_async == (mode == "2")
    for it in range(1, 251):
        tr.train_step(it)                       # (mode 2 reads its flag words at iteration 250: nothing flagged)
    with torch.no_grad():
        model.sh[5, 0, 0] = float("nan")        # (a colour coefficient: the geometry stays what it was)
    with pytest.raises(FloatingPointError) as e:
        for it in range(251, 501):
            tr.train_step(it)
    assert "iteration 251" in str(e.value), str(e.value)


@pytest.mark.gpu
def test_fused_regularisers_equal_the_separate_ones():
    """Trainer(fused_regularisers=True) adds the scale / opacity regulariser gradients inside the render backward's activation kernel;
    False adds them with elementwise ops behind the backward (trainer.cpp:103-127 puts both terms into the loss): same parameters after
    a few iterations, and one call of the operator alone matches the closed form."""
    import gsx  # noqa: F401
    from gsx import ops, parameters, rasterizer, scenes, trainer
    dev = "cuda:0"
    g = torch.Generator().manual_seed(1)
    n = 1000
    sr, rr, orw = torch.randn(n, 3, generator=g).to(dev), torch.randn(n, 4, generator=g).to(dev), torch.randn(n, generator=g).to(dev)
    vs, vq, vo = torch.randn(n, 3, generator=g).to(dev), torch.randn(n, 4, generator=g).to(dev), torch.randn(n, generator=g).to(dev)
    a = ops.splat_activations_bwd(sr, rr, orw, vs, vq, vo)
    b = ops.splat_activations_bwd(sr, rr, orw, vs, vq, vo, None, None, None, 0.01 / sr.numel(), 0.02 / orw.numel())
    sg = torch.sigmoid(orw)
    assert torch.equal(a[1], b[1])
    assert torch.allclose(b[0], a[0] + 0.01 / sr.numel() * torch.exp(sr), rtol=1e-6, atol=1e-9)
    assert torch.allclose(b[2], a[2] + 0.02 / orw.numel() * sg * (1 - sg), rtol=1e-6, atol=1e-9)
    res = []
    for fused in (False, True):
        sc, gt_model, cams = _scene(dev, N=3000, K=16)
        bg = sc["background"].to(dev)
        with torch.no_grad():
            images = [rasterizer.rasterize_fused(c, gt_model, bg).image.clone() for c in cams]
        gg = torch.Generator().manual_seed(9)
        model = scenes.to_splat_data(dict(sc), dev)
        model.sh = (gt_model.sh + 0.3 * torch.randn(gt_model.sh.shape, generator=gg).to(dev)).contiguous()
        prm = parameters.OptimizationParameters(iterations=200, start_refine=100, refine_every=100, stop_refine=150, max_cap=3000, sh_degree_interval=1000)
        assert prm.scale_reg > 0 and prm.opacity_reg > 0
        tr = trainer.Trainer(model, cams, images, prm, bg, seed=3, fused_regularisers=fused)
        assert all(im.is_contiguous() for im in tr.images)   # (the clamped renders above are HWC-strided views: made dense once)
        for it in range(1, 11):
            tr.train_step(it)
        torch.cuda.synchronize()
        res.append(torch.cat([p.detach().reshape(-1) for p in model.params()]))
    rel = float((res[0] - res[1]).norm() / res[0].norm())
    assert rel < 1e-5, rel


@pytest.mark.gpu
def test_train_colmap_example_end_to_end(tmp_path):
    """BASELINE configs[2] as a command (examples/train_colmap.py = the reference's `LichtFeld-Studio -d <capture> --images ... --iter ... --config ...`):
    a capture on disk — binary COLMAP sparse model + PNG images, written by the example's own synthetic-capture writer — goes through the COLMAP
    reader, the image IO, init_model_from_pointcloud, the Trainer (two refine events) and the PLY export; the held-out views must improve."""
    import argparse
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("train_colmap", os.path.join(root, "examples", "train_colmap.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cap = mod.make_synthetic(str(tmp_path / "capture"))
    assert os.path.exists(os.path.join(cap, "sparse", "0", "points3D.bin")) and len(os.listdir(os.path.join(cap, "images"))) == 24
    args = argparse.Namespace(data=cap, images="images", iter=700, config=None, eval=True, test_every=8, resize_factor=-1, max_width=3840,
                              output=str(tmp_path / "out"), json=str(tmp_path / "res.json"), log_every=0)
    res = mod.run(args)
    assert res["iterations"] == 700 and res["gaussians_end"] > res["gaussians_start"]          # two growth steps (iterations 500, 600)
    assert res["psnr_after"] > res["psnr_before"] + 2.0 and res["ssim_after"] > res["ssim_before"], res
    assert os.path.exists(res["ply"]) and os.path.exists(str(tmp_path / "res.json"))
