"""The fused render (rasterize_fused: fused activation / SH-colour glue kernels, one autograd node, optional gradient
sinks) must give the same image and the same parameter gradients as the reference-style chain (rasterize)."""
import numpy as np
import pytest
import torch

from tests.helpers import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def mods():
    import gsx  # noqa: F401
    from gsx import distributed, ops, rasterizer, scenes
    return distributed, ops, rasterizer, scenes


def _setup(scenes, rasterizer, deg):
    sc = scenes.scene_small(seed=17, N=6000)
    sc["width"], sc["height"] = 160, 112
    sc["K"] = scenes.intrinsics(120.0, 120.0, 80.0, 56.0)
    g = torch.Generator().manual_seed(3)
    sc["sh"] = (torch.rand(6000, 16, 3, generator=g) - 0.5) * 0.6
    sc["sh_degree"] = deg
    vm = scenes.look_at_viewmat((0.3, -0.2, -0.5), (0.0, 0.0, 2.5))
    sc["viewmat"] = vm
    cam = rasterizer.Camera(viewmat=vm.to(DEV), K=sc["K"].to(DEV), width=160, height=112)
    return sc, cam


@pytest.mark.parametrize("deg", [0, 2, 3])
def test_fused_equals_unfused(mods, deg):
    distributed, ops, rasterizer, scenes = mods
    sc, cam = _setup(scenes, rasterizer, deg)
    bg = sc["background"].to(DEV) + 0.2
    w = torch.linspace(0.5, 1.5, 160, device=DEV)

    def run(fn, **kw):
        model = scenes.to_splat_data(sc, DEV)
        for p in model.params():
            p.requires_grad_(True)
        out = fn(cam, model, bg, **kw)
        ((out.image * w).sum() + 0.3 * out.alpha.sum()).backward()
        return model, out

    m_ref, o_ref = run(rasterizer.rasterize)
    m_fus, o_fus = run(rasterizer.rasterize_fused)
    assert o_ref.n_isects == o_fus.n_isects
    assert float((o_ref.image - o_fus.image).abs().max()) < 2e-5
    assert float((o_ref.alpha - o_fus.alpha).abs().max()) < 2e-5
    for a, b, n in zip(m_ref.params(), m_fus.params(), ["means", "sh", "scaling_raw", "rotation_raw", "opacity_raw"]):
        e = rel_l2(b.grad.cpu().numpy(), a.grad.cpu().numpy())
        assert e < 1e-3, (n, e)
    # gradient sinks: backward writes into the flat bucket, autograd sees no gradient
    model = scenes.to_splat_data(sc, DEV)
    for p in model.params():
        p.requires_grad_(True)
    bucket = distributed.GradBucket(model.params())
    bucket.flat.fill_(float("nan"))       # every element must be overwritten
    out = rasterizer.rasterize_fused(cam, model, bg, grad_sinks=bucket.sinks())
    ((out.image * w).sum() + 0.3 * out.alpha.sum()).backward()
    assert all(bool(torch.isfinite(p.grad).all()) for p in model.params())   # (the bucket's alignment pads are not written)
    for a, b, n in zip(m_fus.params(), model.params(), ["means", "sh", "scaling_raw", "rotation_raw", "opacity_raw"]):
        assert rel_l2(b.grad.cpu().numpy(), a.grad.cpu().numpy()) < 1e-4, n


def test_fused_keeps_distortion(mods):
    """A distorted COLMAP-style camera: the fused render must apply radial / tangential coefficients like rasterize()
    (rasterizer.cpp:183-195) — and differ from the undistorted render, so the test cannot pass on dropped coefficients."""
    distributed, ops, rasterizer, scenes = mods
    sc, cam = _setup(scenes, rasterizer, 2)
    bg = sc["background"].to(DEV) + 0.2
    cam_d = rasterizer.Camera(viewmat=cam.viewmat, K=cam.K, width=cam.width, height=cam.height,
                              radial=torch.tensor([-0.12, 0.03]), tangential=torch.tensor([0.004, -0.003]))

    def run(fn, c):
        model = scenes.to_splat_data(sc, DEV)
        for p in model.params():
            p.requires_grad_(True)
        out = fn(c, model, bg)
        (out.image.sum() + 0.3 * out.alpha.sum()).backward()
        return model, out

    m_ref, o_ref = run(rasterizer.rasterize, cam_d)
    m_fus, o_fus = run(rasterizer.rasterize_fused, cam_d)
    _, o_plain = run(rasterizer.rasterize_fused, cam)
    assert float((o_fus.image - o_plain.image).abs().max()) > 1e-2      # the coefficients matter
    assert o_ref.n_isects == o_fus.n_isects
    assert float((o_ref.image - o_fus.image).abs().max()) < 2e-5
    for a, b, n in zip(m_ref.params(), m_fus.params(), ["means", "sh", "scaling_raw", "rotation_raw", "opacity_raw"]):
        assert rel_l2(b.grad.cpu().numpy(), a.grad.cpu().numpy()) < 1e-3, n


@pytest.mark.parametrize("camera", ["pinhole", "distorted", "fisheye"])
def test_activation_epilogue_equals_the_separate_launch(mods, camera, monkeypatch):
    """Round 6 (include/gsx.h ABI 7, ops.rasterize_bwd_act): the activation Jacobians applied by the backward's gather kernel against the
    separate splat_activations_bwd launch — the raw-parameter gradients must be the same values (the two runs sum a Gaussian's moment
    records in whatever order its tiles finished: rounding-level differences only), every element of the sinks written (NaN pre-fill,
    Gaussians no tile touches included), the MCMC regulariser terms included.  Fisheye takes the C ABI's second route (the epilogue is
    not folded where a reference-order kernel may add flagged tiles on top): same entry point, same values."""
    distributed, ops, rasterizer, scenes = mods
    sc, cam = _setup(scenes, rasterizer, 2)
    if camera == "distorted":
        cam = rasterizer.Camera(viewmat=cam.viewmat, K=cam.K, width=cam.width, height=cam.height, radial=torch.tensor([-0.12, 0.03]),
                                tangential=torch.tensor([0.004, -0.003]))
    elif camera == "fisheye":
        cam = rasterizer.Camera(viewmat=cam.viewmat, K=cam.K, width=cam.width, height=cam.height, camera_model=ops.CameraModelType.FISHEYE,
                                radial=torch.tensor([0.01, -0.002, 0.0, 0.0]))
    bg = sc["background"].to(DEV) + 0.2
    w = torch.linspace(0.5, 1.5, 160, device=DEV)
    sc = dict(sc)
    sc["quats"] = sc["quats"] * torch.linspace(0.3, 3.0, sc["quats"].shape[0]).unsqueeze(-1)   # raw quaternion norms != 1: the Jacobian needs them
    sc["means"] = sc["means"].clone()
    sc["means"][:100, 2] = -5.0     # behind the camera: no tile touches them
    grads = {}
    for act in (False, True):
        monkeypatch.setattr(rasterizer, "ACT_EPILOGUE", act)
        model = scenes.to_splat_data(sc, DEV)
        for p in model.params():
            p.requires_grad_(True)
        bucket = distributed.GradBucket(model.params())
        bucket.flat.fill_(float("nan"))
        sinks = bucket.sinks()
        sinks["_regularisers"] = (0.01 / model.scaling_raw.numel(), 0.02 / model.opacity_raw.numel())
        out = rasterizer.rasterize_fused(cam, model, bg, grad_sinks=sinks)
        ((out.image * w).sum() + 0.3 * out.alpha.sum()).backward()
        torch.cuda.synchronize()
        grads[act] = {n: getattr(model, n).grad.detach().clone() for n in ("means", "scaling_raw", "rotation_raw", "opacity_raw")}
        assert all(bool(torch.isfinite(g).all()) for g in grads[act].values()), (camera, act)
    untouched = (grads[True]["means"] == 0).all(-1)
    assert int(untouched.sum()) > 50     # the scene has Gaussians outside the view: their regulariser terms must still arrive
    assert float(grads[True]["scaling_raw"][untouched].abs().min()) > 0 and float(grads[True]["opacity_raw"].reshape(-1)[untouched].abs().min()) > 0
    for n in grads[True]:
        a, b = grads[False][n], grads[True][n]
        assert rel_l2(b.cpu().numpy(), a.cpu().numpy()) < 2e-6, (camera, n)
        assert torch.equal(a[untouched], b[untouched]), (camera, n)   # no record sums involved: bit for bit


def test_fused_ops_vs_torch(mods):
    """The two fused glue ops against the torch expressions they replace."""
    _, ops, _, _ = mods
    g = torch.Generator(device=DEV).manual_seed(0)
    N = 5000
    sr = torch.randn(N, 3, device=DEV, generator=g) * 0.5 - 3
    rr = torch.randn(N, 4, device=DEV, generator=g)
    orw = torch.randn(N, device=DEV, generator=g)
    s, q, o = ops.splat_activations_fwd(sr, rr, orw)
    assert torch.allclose(s, torch.exp(sr), rtol=1e-6, atol=0) and torch.allclose(o, torch.sigmoid(orw), rtol=1e-6, atol=1e-7)
    assert torch.allclose(q, torch.nn.functional.normalize(rr, dim=-1), rtol=1e-6, atol=1e-7)
    sr2, rr2, or2 = (t.clone().requires_grad_(True) for t in (sr, rr, orw))
    vs, vq, vo = torch.randn_like(sr), torch.randn_like(rr), torch.randn_like(orw)
    (torch.exp(sr2) * vs).sum().backward(); (torch.nn.functional.normalize(rr2, dim=-1) * vq).sum().backward(); (torch.sigmoid(or2) * vo).sum().backward()
    gs, gr, go = ops.splat_activations_bwd(sr, rr, orw, vs, vq, vo, None, None, None)
    assert torch.allclose(gs, sr2.grad, rtol=1e-5, atol=1e-7) and torch.allclose(gr, rr2.grad, rtol=1e-4, atol=1e-6)
    assert torch.allclose(go, or2.grad, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("iteration,deg", [(1500, 3), (500, 3), (1500, 1)])
def test_sh_backward_fused_with_adam_equals_separate_step(mods, iteration, deg):
    """Training steps through rasterize_fused with the SH tensor's Adam step applied inside the SH backward (gsx_sh_colors_bwd_adam,
    optim.FusedAdam.begin_fused_sh_step) against the same steps with the SH gradient written and optimizer.step() over all six groups:
    parameters, moments and step counters must agree — including the shN warm-up quirk (iteration <= 1000: shN untouched, counter
    advances) and the rows beyond the active degree (zero gradient, still stepped)."""
    distributed, ops, rasterizer, scenes = mods
    from gsx import optim
    sc, cam = _setup(scenes, rasterizer, deg)
    bg = sc["background"].to(DEV)
    target = torch.rand(1, 112, 160, 3, generator=torch.Generator().manual_seed(5)).to(DEV)
    results = []
    for fused in (False, True):
        model = scenes.to_splat_data(dict(sc), DEV)
        for p in model.params():
            p.requires_grad_(True)
        bucket = distributed.GradBucket(model.params())
        sinks = bucket.sinks()
        opt = optim.FusedAdam.for_splat_data(model)
        for it in range(iteration, iteration + 3):
            sinks["_sh_adam"] = opt.begin_fused_sh_step(it) if fused else None
            assert (sinks["_sh_adam"] is not None) == fused
            out = rasterizer.rasterize_fused(cam, model, bg, grad_sinks=sinks)
            ((out.render_hwc - target) ** 2).mean().backward()
            opt.step(it, skip_sh=fused)
        torch.cuda.synchronize()
        st = opt.state["sh"]
        results.append(dict(sh=model.sh.detach().clone(), means=model.means.detach().clone(), m=st["exp_avg"].clone(), v=st["exp_avg_sq"].clone(),
                            v_means=opt.state["means"]["exp_avg_sq"].clone(),
                            steps=(opt.step_count("sh0"), opt.step_count("shN"), opt.step_count("means"))))
    a, b = results
    assert a["steps"] == b["steps"] == (3, 3, 3)
    for k in ("sh", "m", "v", "means"):
        assert torch.isfinite(b[k]).all()
        # same arithmetic in two kernels (FMA contraction may differ), and the two runs sum the backward's moment records of a Gaussian in
        # whatever order its tiles finished: the GRADIENTS agree to fp32 summation noise — a few 1e-6 of the tensor's largest
        # gradient — and Adam turns a gradient perturbation dg into an update perturbation of lr * dg / |g| per step: nothing for an element
        # with a real gradient, a full +-lr for one whose gradient IS that noise (either sign can come out; seen once in ~20 runs of the
        # suite).  ADVICE r05: the bound therefore follows the element's gradient magnitude — its second moment, which noise does not
        # amplify and which is compared strictly below — instead of allowing a few outliers anywhere: a sign or indexing bug on an element
        # with a real gradient (sqrt(v) >= 1e-2 of the largest) has 1.5e-6 of slack, not +-lr.
        diff = (a[k] - b[k]).abs()
        base = 1e-5 * float(a[k].abs().max()) + 1e-12
        if k in ("sh", "means"):
            lr = 2.5e-3 if k == "sh" else 1.6e-4
            g = (a["v"] if k == "sh" else a["v_means"]).sqrt()
            amp = (2e-6 * float(g.max()) / g.clamp_min(1e-30)).clamp(max=2.1)      # |dg| / |g|, capped at "opposite signs"
            bad = diff > base + 3 * lr * amp
            assert not bool(bad.any()), (k, int(bad.sum()), (g[bad] / g.max()).tolist()[:5], diff[bad].tolist()[:5], float(a[k].abs().max()))
            assert int((diff > base).sum()) <= 10, (k, int((diff > base).sum()))
        else:
            assert not bool((diff > base).any()), (k, float(diff.max()))
    if iteration <= 1000:   # shN frozen: its block is untouched in both
        assert torch.equal(b["sh"][:, 1:], scenes.to_splat_data(dict(sc), DEV).sh[:, 1:])
    assert float((b["sh"][:, :1] - scenes.to_splat_data(dict(sc), DEV).sh[:, :1]).abs().max()) > 0


def test_activations_projection_in_one_launch_is_bit_identical(mods):
    """gsx_splat_activations_projection_ut == gsx_splat_activations_fwd followed by gsx_projection_ut_3dgs_fused, bit for bit (every
    output, incl. the culled Gaussians' radii), for a perfect pinhole and a distorted one."""
    distributed, ops, rasterizer, scenes = mods
    sc, cam = _setup(scenes, rasterizer, 3)
    model = scenes.to_splat_data(dict(sc), DEV)
    ut = ops.UnscentedTransformParameters()
    vm, K = cam.world_view_transform().contiguous(), cam.K_batched().contiguous()
    sr, rr, orw = model.scaling_raw.contiguous(), model.rotation_raw.contiguous(), model.opacity_raw.reshape(-1).contiguous()
    for radial in (None, torch.tensor([0.05, -0.01, 0.0, 0.0, 0.0, 0.0], device=DEV)):
        s1, q1, o1 = ops.splat_activations_fwd(sr, rr, orw)
        r1, m1, d1, c1, _ = ops.projection_ut_3dgs_fused(model.means, q1, s1, o1, vm, None, K, 160, 112, 0.3, 0.01, 10000.0, 0.0, False,
                                                        ops.CameraModelType.PINHOLE, ut, ops.ShutterType.GLOBAL, radial, None, None)
        s2, q2, o2, r2, m2, d2, c2 = ops.splat_activations_projection_ut(model.means, sr, rr, orw, vm, K, 160, 112, 0.3, 0.01, 10000.0, 0.0,
                                                                       ops.CameraModelType.PINHOLE, ut, radial, None, None)
        assert torch.equal(s1, s2) and torch.equal(q1, q2) and torch.equal(o1, o2) and torch.equal(r1, r2)
        vis = (r1 > 0).all(-1)
        assert int(vis.sum()) > 1000
        for a, b in ((m1, m2), (d1, d2), (c1, c2)):   # outputs of culled Gaussians are not written by either
            assert torch.equal(a[vis], b[vis])


@pytest.mark.parametrize("deg,distorted", [(3, False), (0, False), (2, True)])
def test_frontend_fused_is_bit_identical_to_the_separate_operators(mods, deg, distorted):
    """ops.frontend_fused (activations -> UT projection -> SH colours -> packed blend records in ONE kernel, csrc/gsx_frontend.hip) against the
    launches it replaces: activated parameters, projection and colours bit for bit; the packed records to the last bit or two (see
    below), hence the blend forward / backward on them equal to rounding to the ones that pack for themselves."""
    distributed, ops, rasterizer, scenes = mods
    sc, cam = _setup(scenes, rasterizer, deg)
    model = scenes.to_splat_data(sc, DEV)
    ut = ops.UnscentedTransformParameters()
    vm, K = cam.world_view_transform().contiguous(), cam.K_batched().contiguous()
    radial = torch.tensor([[0.05, -0.02, 0.003, 0.0, 0.0, 0.0]], device=DEV) if distorted else None
    tang = torch.tensor([[0.002, -0.001]], device=DEV) if distorted else None
    sr, rr, orw = model.scaling_raw.contiguous(), model.rotation_raw.contiguous(), model.opacity_raw.reshape(-1).contiguous()
    W, H = 96, 64      # a crop of the camera's image (principal point off centre): part of the scene is culled
    cm = ops.CameraModelType.PINHOLE
    fe = ops.frontend_fused(deg, model.means, model.sh, sr, rr, orw, vm, K, W, H, 0.3, 0.01, 1e4, 0.0, cm, ut, radial, tang, None)
    assert fe[8] is not None
    scales, quats, opac, radii, means2d, depths, conics, colors, ws = fe
    ref = ops.splat_activations_projection_ut(model.means, sr, rr, orw, vm, K, W, H, 0.3, 0.01, 1e4, 0.0, cm, ut, radial, tang, None)
    col_ref = ops.sh_colors_fwd(deg, model.means, vm, model.sh, ref[3])
    vis = (ref[3] > 0).all(-1)
    assert 0.2 < float(vis.float().mean()) < 1.0   # the camera culls some
    for a, b, n in zip((scales, quats, opac, radii), ref[:4], ("scales", "quats", "opacities", "radii")):
        assert torch.equal(a, b), n
    for a, b, n in zip((means2d, depths, conics), ref[4:], ("means2d", "depths", "conics")):
        assert torch.equal(a[vis], b[vis]), n       # (only radii is written for a culled Gaussian)
    assert torch.equal(colors, col_ref)
    tw, th = (W + 15) // 16, (H + 15) // 16
    _, _, fl, off = ops.intersect_tile_binned(means2d, radii, depths, 1, 16, tw, th, False)
    bg = sc["background"][None].to(DEV)
    args = (model.means, quats, scales, colors, opac[None].contiguous(), bg, None, W, H, 16, vm, None, K, cm, ut, ops.ShutterType.GLOBAL, radial, tang, None, off, fl)
    a = ops.rasterize_fwd_packed(*args, ws)
    b = ops.rasterize_to_pixels_from_world_3dgs_fwd(*args, keep_ws=True)
    # the packed records: the same device function compiled into two kernels — the compiler contracts a few of its a*b+c differently, so
    # five of the sixteen fields may differ in the last bit; everything downstream agrees to rounding
    def recs(w):
        base = (w.data_ptr() + 255) // 256 * 256 - w.data_ptr()
        return w[base:base + model.means.shape[0] * 64].view(torch.float32).reshape(-1, 16)
    r_fe, r_pk = recs(ws)[vis[0]], recs(b[3])[vis[0]]
    # fields 0..13; 14 / 15 carry the Gaussian's rectangle of 16-px tiles (front end: from means2d / radii; pack kernel: "everything")
    assert float(((r_fe[:, :14] - r_pk[:, :14]).abs() / r_pk[:, :14].abs().amax(1, keepdim=True)).max()) < 5e-7   # relative to the record's largest entry
    rect = r_fe[:, 14:].contiguous().view(torch.int32)
    x0, x1, y0, y1 = rect[:, 0] & 0xFFFF, (rect[:, 0] >> 16) & 0xFFFF, rect[:, 1] & 0xFFFF, (rect[:, 1] >> 16) & 0xFFFF
    m2, rd = means2d[0][vis[0]], radii[0][vis[0]].float()
    ex0 = torch.clamp(torch.floor(m2[:, 0] / 16.0 - rd[:, 0] / 16.0), 0, tw).int()
    ex1 = torch.clamp(torch.ceil(m2[:, 0] / 16.0 + rd[:, 0] / 16.0), 0, tw).int()
    ey0 = torch.clamp(torch.floor(m2[:, 1] / 16.0 - rd[:, 1] / 16.0), 0, th).int()
    ey1 = torch.clamp(torch.ceil(m2[:, 1] / 16.0 + rd[:, 1] / 16.0), 0, th).int()
    assert torch.equal(x0, ex0) and torch.equal(x1, ex1) and torch.equal(y0, ey0) and torch.equal(y1, ey1)   # IntersectTile.cu:65-76
    # the invariant the RANGES layout of the backward's records stands on: the rectangles of the visible Gaussians sum to n_isects
    assert int(((x1 - x0).long() * (y1 - y0).long()).sum()) == int(fl.numel())
    assert bool((r_pk[:, 14:].contiguous().view(torch.int32) == -65536).all())                                # 0xFFFF0000: no restriction
    # a Gaussian the projection culled gets a NULL record (log2 opacity = -inf: alpha 0 everywhere), never uninitialised memory
    r_cull = recs(ws)[~vis[0]]
    assert r_cull.shape[0] > 0 and bool(torch.isneginf(r_cull[:, 5]).all()) and bool((r_cull[:, 14:] == 0).all())
    assert float((a[0] - b[0]).abs().max()) < 2e-6 and float((a[1] - b[1]).abs().max()) < 2e-6 and torch.equal(a[2], b[2])
    g = torch.Generator().manual_seed(1)
    v_rc, v_ra = torch.randn(1, H, W, 3, generator=g).to(DEV), torch.randn(1, H, W, 1, generator=g).to(DEV)
    ga = ops.rasterize_to_pixels_from_world_3dgs_bwd(*args, a[1], a[2], v_rc, v_ra, fwd_ws=ws)
    gb = ops.rasterize_to_pixels_from_world_3dgs_bwd(*args, a[1], a[2], v_rc, v_ra)
    assert all(rel_l2(x.cpu().numpy(), y.cpu().numpy()) < 1e-5 for x, y in zip(ga, gb))
    # the render path's variant with the backward's records in RANGES (one contiguous run of slots per Gaussian, sized by its tile rectangle:
    # gsx_raster_common.hpp) instead of chains: the same gradients, and a second backward on the same workspace (the gather has put every
    # count back to 0) repeats the first
    if not distorted:
        fe_r = ops.frontend_fused_render(deg, model.means, model.sh, sr, rr, orw, vm, K, W, H, 0.3, 0.01, 1e4, 0.0, cm, ut, radial, tang, None, True)
        ws_r = fe_r[8]
        a_r = ops.rasterize_fwd_packed(*args, ws_r)
        assert torch.equal(a_r[0], a[0]) and torch.equal(a_r[2], a[2])
        gr1 = ops.rasterize_to_pixels_from_world_3dgs_bwd(*args, a_r[1], a_r[2], v_rc, v_ra, fwd_ws=ws_r)
        gr2 = ops.rasterize_to_pixels_from_world_3dgs_bwd(*args, a_r[1], a_r[2], v_rc, v_ra, fwd_ws=ws_r)
        for x, y, z in zip(gr1, gr2, ga):
            assert rel_l2(x.cpu().numpy(), z.cpu().numpy()) < 1e-5 and rel_l2(y.cpu().numpy(), x.cpu().numpy()) < 1e-5
    # a camera the front end does not take: undefined workspace, the caller falls back
    fish = ops.frontend_fused(deg, model.means, model.sh, sr, rr, orw, vm, K, W, H, 0.3, 0.01, 1e4, 0.0, ops.CameraModelType.FISHEYE, ut, None, None, None)
    assert fish[8] is None


@pytest.mark.parametrize("size", [(160, 112), (150, 100)])
def test_lists_per_32px_tiles_give_the_same_render(mods, size, monkeypatch):
    """rasterize_fused with the intersection lists built per 32 x 32 pixels (one list per 2 x 2 pixel tiles: include/gsx.h, the blend entry
    points' tile_size = 32) against the reference's 16-pixel lists: the same Gaussians reach every pixel in the same order, so the image is
    bit-identical; the gradients agree to rounding (the backward's records are summed in another order).  Also an image size that is not a
    multiple of 32."""
    distributed, ops, rasterizer, scenes = mods
    sc, cam = _setup(scenes, rasterizer, 3)
    W, H = size
    cam = rasterizer.Camera(viewmat=cam.viewmat, K=cam.K, width=W, height=H)
    bg = sc["background"].to(DEV) + 0.1
    w = torch.linspace(0.5, 1.5, W, device=DEV)

    def run(list_tile):
        monkeypatch.setenv("GSX_LIST_TILE", str(list_tile))
        model = scenes.to_splat_data(sc, DEV)
        for p in model.params():
            p.requires_grad_(True)
        out = rasterizer.rasterize_fused(cam, model, bg)
        ((out.image * w).sum() + 0.3 * out.alpha.sum()).backward()
        return model, out

    m16, o16 = run(16)
    m32, o32 = run(32)
    assert o32.n_isects < o16.n_isects                      # fewer keys ...
    assert torch.equal(o16.image, o32.image) and torch.equal(o16.alpha, o32.alpha)   # ... the same image, bit for bit
    for a, b, n in zip(m16.params(), m32.params(), ["means", "sh", "scaling_raw", "rotation_raw", "opacity_raw"]):
        assert rel_l2(b.grad.cpu().numpy(), a.grad.cpu().numpy()) < 1e-5, n   # (two launches of the backward differ by their record order: ~1e-6)
    # the hysteresis rule: dense lists switch a problem shape to 32-pixel lists, sparse ones back
    monkeypatch.delenv("GSX_LIST_TILE")
    key = ("shape",)
    rasterizer._LIST_TILE_STATE.pop(key, None)
    assert rasterizer._list_tile_for(key) == 16
    rasterizer._list_tile_update(key, 16, 4000 * 100, 100)
    assert rasterizer._list_tile_for(key) == 32
    rasterizer._list_tile_update(key, 32, 9000 * 25, 25)
    assert rasterizer._list_tile_for(key) == 32
    rasterizer._list_tile_update(key, 32, 1000 * 25, 25)
    assert rasterizer._list_tile_for(key) == 16
    # no flipping: a frame that just switched up (3 200 keys per 16-px tile = ~4 100 per 32-px tile of the same frame) stays up
    rasterizer._list_tile_update(key, 16, 3200 * 400, 400)
    assert rasterizer._list_tile_for(key) == 32
    rasterizer._list_tile_update(key, 32, int(3200 * 400 / 3.1), 100)
    assert rasterizer._list_tile_for(key) == 32


def test_lists_per_32px_tiles_respect_the_16px_rectangles(mods, monkeypatch):
    """with 32-pixel lists a 16-pixel tile walks its parent's list, and the reference composites a Gaussian only in
    the 16-pixel tiles of ITS rectangle (means2d +- radii, IntersectTile.cu:65-76) — whatever alpha the 3-D evaluation gives elsewhere.
    Long, near, oblique needles are where the UT-projected rectangle under-covers the ray-space footprint: the packed record carries the
    rectangle and staging drops the entries that do not contain the tile.  Same image bit for bit, and the same as the unfused chain."""
    distributed, ops, rasterizer, scenes = mods
    N, W, H = 400, 192, 128
    g = torch.Generator().manual_seed(11)
    sc = scenes.scene_small(seed=4, N=N)
    sc["means"] = torch.cat([(torch.rand(N, 2, generator=g) - 0.5) * 1.2, 0.4 + 1.6 * torch.rand(N, 1, generator=g)], 1)
    sc["scales"] = torch.stack([0.25 + 0.5 * torch.rand(N, generator=g), 0.004 + 0.01 * torch.rand(N, generator=g), 0.004 + 0.01 * torch.rand(N, generator=g)], 1)
    sc["quats"] = torch.nn.functional.normalize(torch.randn(N, 4, generator=g), dim=-1)
    sc["opacities"] = 0.6 + 0.39 * torch.rand(N, generator=g)
    sc["sh"] = (torch.rand(N, 16, 3, generator=g) - 0.5) * 0.6
    sc["sh_degree"] = 3
    sc["width"], sc["height"] = W, H
    sc["K"] = scenes.intrinsics(110.0, 110.0, W / 2.0, H / 2.0)
    cam = rasterizer.Camera(viewmat=torch.eye(4, device=DEV), K=sc["K"].to(DEV), width=W, height=H)
    bg = sc["background"].to(DEV) + 0.1
    wgt = torch.linspace(0.5, 1.5, W, device=DEV)

    def run(list_tile, fn):
        monkeypatch.setenv("GSX_LIST_TILE", str(list_tile))
        model = scenes.to_splat_data(sc, DEV)
        for p in model.params():
            p.requires_grad_(True)
        out = fn(cam, model, bg)
        ((out.image * wgt).sum() + 0.3 * out.alpha.sum()).backward()
        return model, out

    m16, o16 = run(16, rasterizer.rasterize_fused)
    m32, o32 = run(32, rasterizer.rasterize_fused)
    mref, oref = run(16, rasterizer.rasterize)
    # the scene does what it was built for: Gaussians whose alpha reaches pixels OUTSIDE their rectangle of 16-px tiles exist
    # (checked through the records: entries of 32-px lists dropped by the rectangle although the footprint test lets them through
    # would change the image if the filter were missing — asserted indirectly by comparing with a render that has no such filter to apply)
    assert o32.n_isects < o16.n_isects
    assert torch.equal(o16.image, o32.image) and torch.equal(o16.alpha, o32.alpha)
    assert float((oref.image - o32.image).abs().max()) < 2e-5
    for a, b, n in zip(m16.params(), m32.params(), ["means", "sh", "scaling_raw", "rotation_raw", "opacity_raw"]):
        assert rel_l2(b.grad.cpu().numpy(), a.grad.cpu().numpy()) < 1e-4, n   # (needles: the record order moves more bits than in the test above)
    # ... and the scene does exercise the case: without the rectangle filter (test switch) the 32-px lists composite extra pairs
    monkeypatch.setenv("GSX_LIST_RECT", "0")
    _, o32_nofilter = run(32, rasterizer.rasterize_fused)
    monkeypatch.delenv("GSX_LIST_RECT")
    n_diff = int((o32_nofilter.image != o16.image).any(0).sum())
    print("pixels that differ without the 16-px rectangle filter:", n_diff, "max", float((o32_nofilter.image - o16.image).abs().max()))
    assert n_diff > 0, "the needle scene no longer has footprints that leave their 16-px rectangles: the test above proves nothing"


@pytest.mark.parametrize("size,list_tile", [((160, 112), 16), ((150, 100), 16), ((131, 77), 16), ((150, 100), 32)])
def test_forward_kernels_are_bit_identical(mods, size, list_tile, monkeypatch):
    """The three forward kernels of the fast path (GSX_FWD=wave: one list per 8x8 quadrant; quad: four lists per wave, one per DPP row / 4x4
    block; pair: two pixels per lane, eight lists per wave) evaluate the same pairs in the same order with the same instructions: image, alpha
    and last ids are EQUAL — ragged image sizes, 16- and 32-pixel lists; and the launcher's own choice (no switch) is one of them."""
    distributed, ops, rasterizer, scenes = mods
    sc, cam = _setup(scenes, rasterizer, 3)
    W, H = size
    cam = rasterizer.Camera(viewmat=cam.viewmat, K=cam.K, width=W, height=H)
    bg = sc["background"].to(DEV) + 0.1
    monkeypatch.setenv("GSX_LIST_TILE", str(list_tile))
    outs = {}
    for mode in ("wave", "quad", "pair", None):
        if mode is None:
            monkeypatch.delenv("GSX_FWD")
        else:
            monkeypatch.setenv("GSX_FWD", mode)
        model = scenes.to_splat_data(sc, DEV)
        with torch.no_grad():
            o = rasterizer.rasterize_fused(cam, model, bg)
        outs[mode] = (o.image.clone(), o.alpha.clone())
    for k in ("quad", "pair", None):
        assert torch.equal(outs["wave"][0], outs[k][0]) and torch.equal(outs["wave"][1], outs[k][1]), k
