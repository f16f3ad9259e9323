"""Guarded intersection lists (include/gsx.h "guarded lists", rasterize_fused(guarded=True)): the render path without a host read of
n_isects.  Same image and gradients as the exact protocol bit for bit while the capacity suffices; a frame that outgrows it renders
EMPTY lists on the device, is reported before anything irreversible ran, and renders correctly when repeated."""
import ctypes
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mods():
    import gsx  # noqa: F401
    from gsx import distributed, loss, ops, optim, rasterizer, scenes
    return distributed, loss, ops, optim, rasterizer, scenes


def _scene(scenes, rasterizer, N, W, H, eye):
    sc = scenes.scene_small(seed=23, N=N)
    sc["width"], sc["height"] = W, H
    sc["K"] = scenes.intrinsics(0.75 * W, 0.75 * W, W / 2.0, H / 2.0)
    g = torch.Generator().manual_seed(5)
    sc["sh"] = (torch.rand(N, 16, 3, generator=g) - 0.5) * 0.6
    sc["sh_degree"] = 3
    vm = scenes.look_at_viewmat(eye, (0.0, 0.0, 2.5))
    return sc, rasterizer.Camera(viewmat=vm.to(DEV), K=sc["K"].to(DEV), width=W, height=H)


def _render(rasterizer, scenes, distributed, sc, cam, guarded, w):
    model = scenes.to_splat_data(sc, DEV)
    for p in model.params():
        p.requires_grad_(True)
    bucket = distributed.GradBucket(model.params())
    bucket.flat.fill_(float("nan"))
    bg = sc["background"].to(DEV) + 0.15
    out = rasterizer.rasterize_fused(cam, model, bg, grad_sinks=bucket.sinks(), guarded=guarded)
    ((out.render_hwc.squeeze(0) * w[None, :, None]).sum() + 0.3 * out.alpha.sum()).backward()
    return out, bucket


def test_guarded_equals_exact_bitwise(mods):
    distributed, loss, ops, optim, rasterizer, scenes = mods
    W, H = 208, 144   # a shape no other test uses: the capacity hint of this problem shape starts cold here
    sc, cam = _scene(scenes, rasterizer, 7000, W, H, (0.2, -0.1, -0.4))
    w = torch.linspace(0.5, 1.5, W, device=DEV)
    o_exact, b_exact = _render(rasterizer, scenes, distributed, sc, cam, False, w)
    assert o_exact.lists is None
    ops.shim_guarded_stats(True)
    o_g, b_g = _render(rasterizer, scenes, distributed, sc, cam, True, w)   # warm hint: the guarded protocol proper
    calls, waits, misses = ops.shim_guarded_stats(True)
    assert calls == 1 and misses == 0
    assert o_g.lists is not None and o_g.lists.status is not None, "the guarded protocol did not run (cold hint?)"
    assert o_g.confirm() and o_g.n_isects == o_exact.n_isects
    assert int(o_g.aux["flatten_ids"].shape[0]) >= o_g.n_isects            # capacity length, not narrowed
    assert int(o_g.lists.status.item()) == o_exact.n_isects                # the device-side verdict = the total
    n = o_exact.n_isects
    assert torch.equal(o_g.aux["flatten_ids"][:n], o_exact.aux["flatten_ids"])
    assert torch.equal(o_g.render_hwc, o_exact.render_hwc) and torch.equal(o_g.alpha, o_exact.alpha)
    # gradients: same kernels, same lists -> same bits up to the backward's launch-order rounding (chained records), which is ~1e-7
    fe, fg = b_exact.flat, b_g.flat
    ok = torch.isfinite(fe)
    assert torch.equal(ok, torch.isfinite(fg))
    rel = float((fe[ok] - fg[ok]).norm() / fe[ok].norm())
    assert rel < 1e-5, rel


def test_overflow_renders_empty_and_is_reported_before_the_optimizer(mods):
    distributed, loss, ops, optim, rasterizer, scenes = mods
    W, H = 224, 160
    N = 9000
    # a distant view first (few intersections: a small capacity hint), then a close view of the same problem shape
    sc_far, cam_far = _scene(scenes, rasterizer, N, W, H, (0.0, 0.0, -14.0))
    sc, cam = _scene(scenes, rasterizer, N, W, H, (0.1, 0.0, -0.2))
    w = torch.linspace(0.5, 1.5, W, device=DEV)
    o_far, _ = _render(rasterizer, scenes, distributed, sc_far, cam_far, False, w)
    model = scenes.to_splat_data(sc, DEV)
    for p in model.params():
        p.requires_grad_(True)
    bucket = distributed.GradBucket(model.params())
    sinks = bucket.sinks()
    opt = optim.FusedAdam.for_splat_data(model)
    sinks["_sh_adam"] = opt.begin_fused_sh_step(1500)
    assert sinks["_sh_adam"] is not None
    sh_before = model.sh.detach().clone()
    bg = sc["background"].to(DEV) + 0.15
    ops.shim_guarded_stats(True)
    out = rasterizer.rasterize_fused(cam, model, bg, grad_sinks=sinks, guarded=True)
    n_true = None
    with pytest.raises(rasterizer.IsectCapacityMiss):
        (out.render_hwc.sum() + out.alpha.sum()).backward()
    n_true, _, complete = out.lists.confirm()
    assert not complete and not out.confirm() and n_true > out.lists.capacity > o_far.n_isects
    assert int(out.lists.status.item()) == -1
    # the overflowed frame rendered empty lists: background everywhere, alpha 0 — and nothing irreversible ran
    assert float((out.render_hwc - bg.reshape(1, 1, 1, 3)).abs().max()) == 0.0 and float(out.alpha.abs().max()) == 0.0
    assert torch.equal(model.sh.detach(), sh_before), "the SH tensor's fused Adam step ran on an overflowed frame"
    assert ops.shim_guarded_stats(True)[2] == 1
    # the repeat (the hint has been raised) is the exact render
    sinks["_sh_adam"] = None
    out2 = rasterizer.rasterize_fused(cam, model, bg, grad_sinks=sinks, guarded=True)
    (out2.render_hwc.sum() + out2.alpha.sum()).backward()
    assert out2.confirm() and out2.n_isects == n_true
    m3 = scenes.to_splat_data(sc, DEV)
    out3 = rasterizer.rasterize_fused(cam, m3, bg)
    assert out3.n_isects == n_true and torch.equal(out3.render_hwc, out2.render_hwc)


def test_training_loop_guarded_matches_exact(mods):
    """A few training iterations over changing cameras: guarded and exact protocols leave the same parameters."""
    distributed, loss, ops, optim, rasterizer, scenes = mods
    from gsx import trainer
    from gsx.strategy import OptimizationParameters
    W, H, N = 176, 128, 5000
    eyes = [(0.3, 0.0, -0.5), (-0.3, 0.1, -0.4), (0.0, -0.3, -6.0), (0.1, 0.2, -0.3), (0.0, 0.0, -0.1)]
    results = []
    for guarded in (False, True):
        sc, _ = _scene(scenes, rasterizer, N, W, H, eyes[0])
        cams = [_scene(scenes, rasterizer, N, W, H, e)[1] for e in eyes]
        g = torch.Generator().manual_seed(9)
        images = [torch.rand(3, H, W, generator=g).to(DEV) for _ in eyes]
        model = scenes.to_splat_data(sc, DEV)
        params = OptimizationParameters()
        params.iterations = 1000
        tr = trainer.Trainer(model, cams, images, params, background=sc["background"].to(DEV), seed=3, guarded_lists=guarded)
        for it in range(1, 21):
            tr.train_step(it)
        torch.cuda.synchronize()
        results.append(([p.detach().clone() for p in model.params()], tr.capacity_misses))
    (pe, _), (pg, misses) = results
    # (a dropped or doubled iteration would move sh / opacity by ~1e-2 of their norm; the backward's launch-order rounding moves them by ~1e-6)
    for a, b in zip(pe, pg):
        assert float((a - b).norm() / a.norm()) < 1e-3
    assert misses <= 3   # (the close-up after the distant view may exceed the capacity once or twice; each is repeated, not lost)


def test_c_abi_status_word(mods):
    """gsx_intersect_bin_count_guarded through the raw C ABI: the device word is the total when capacity and segment bound hold,
    -1 when either is exceeded."""
    distributed, loss, ops, optim, rasterizer, scenes = mods
    c = ctypes
    lib = c.CDLL(os.path.join(ROOT, "gaussian-splatting-cuda_amd", "libgsx.so"))
    lib.gsx_intersect_bin_count_workspace_bytes.restype = c.c_size_t
    N, W, H = 4000, 160, 96
    g = torch.Generator().manual_seed(2)
    means2d = (torch.rand(1, N, 2, generator=g) * torch.tensor([W, H])).to(DEV).contiguous()
    radii = torch.randint(1, 12, (1, N, 2), generator=g, dtype=torch.int32).to(DEV).contiguous()
    tw, th = (W + 15) // 16, (H + 15) // 16
    wsb = lib.gsx_intersect_bin_count_workspace_bytes(c.c_uint32(1), c.c_uint32(tw), c.c_uint32(th))
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    tpg = torch.empty(1, N, dtype=torch.int32, device=DEV)
    off = torch.empty(tw * th + 1, dtype=torch.int32, device=DEV)
    status = torch.full((1,), 12345, dtype=torch.int32, device=DEV)
    host = torch.zeros(1, dtype=torch.int64).pin_memory()

    def run(cap, seg):
        rc = lib.gsx_intersect_bin_count_guarded(c.c_uint32(1), c.c_uint32(N), c.c_void_p(means2d.data_ptr()), c.c_void_p(radii.data_ptr()),
                                                 c.c_uint32(16), c.c_uint32(tw), c.c_uint32(th), c.c_void_p(tpg.data_ptr()),
                                                 c.c_void_p(off.data_ptr()), c.c_void_p(host.data_ptr()), c.c_void_p(ws.data_ptr()),
                                                 c.c_size_t(wsb), c.c_int64(cap), c.c_int64(seg), c.c_void_p(status.data_ptr()), None)
        assert rc == 0
        torch.cuda.synchronize()
        return int(status.item())

    total = int(tpg.sum().item()) if run(1 << 30, 0) >= 0 else -1
    word = int(host.item())
    assert total == int(tpg.sum().item()) == (word & 0xFFFFFFFF) == int(off[-1].item())
    max_seg = word >> 32
    counts = np.diff(off.cpu().numpy())
    assert max_seg == counts.max()
    assert run(total, 0) == total and run(total - 1, 0) == -1
    assert run(total, max_seg) == total and run(total, max_seg - 1) == -1
