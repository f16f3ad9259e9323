"""GPU parity for the camera variants of the path: OpenCV-distorted pinhole and global-shutter fisheye (fast Delta-form path; a
fisheye wider than ~165 degrees mixes it with the reference-order kernels tile by tile), rolling shutter (generic reference-order
path), and C = 2 cameras in one call.  Oracle = oracle/gsx_oracle.cpp."""
import numpy as np
import pytest
import torch

from oracle import oracle
from tests.helpers import np32, oracle_pipeline, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.fixture(autouse=True, params=["wave", "quad"])
def forward_kernel(request, monkeypatch):
    """Every camera case runs with each forward kernel of the fast path forced (one list per 8x8 quadrant / four lists per wave)."""
    monkeypatch.setenv("GSX_FWD", request.param)
    return request.param


@pytest.fixture(scope="module")
def mods():
    import gsx  # noqa: F401
    from gsx import ops, scenes
    return ops, scenes


def _scene(scenes, N=4000, size=128, seed=13, f=90.0):
    sc = scenes.scene_small(seed=seed, N=N)
    sc["width"] = sc["height"] = size
    sc["K"] = scenes.intrinsics(f, f, size / 2.0, size / 2.0)
    sc["background"] = torch.tensor([0.05, 0.1, 0.15])
    return sc


def _run_gpu(ops, sc, o, cam_model, shutter, viewmats1, radial, tangential, thin_prism, v_rc, v_ra):
    args_cam = (t(sc["viewmat"][None].numpy()), None if viewmats1 is None else t(np.asarray(viewmats1, np.float32)),
                t(sc["K"][None].numpy()))
    dist = tuple(None if x is None else t(np.asarray(x, np.float32)) for x in (radial, tangential, thin_prism))
    ut = ops.UnscentedTransformParameters()
    W, H = sc["width"], sc["height"]
    proj = ops.projection_ut_3dgs_fused(sc["means"].to(DEV), sc["quats"].to(DEV), sc["scales"].to(DEV), sc["opacities"].to(DEV),
                                        args_cam[0], args_cam[1], args_cam[2], W, H, 0.3, 0.01, 1e4, 0.0, False, cam_model, ut,
                                        shutter, *dist)
    means, quats, scales = sc["means"].to(DEV), sc["quats"].to(DEV), sc["scales"].to(DEV)
    colors, opac, bg = t(o["colors"]), sc["opacities"][None].to(DEV), sc["background"][None].to(DEV)
    off, fl = t(o["tile_offsets"]), t(o["flatten_ids"])
    fwd = ops.rasterize_to_pixels_from_world_3dgs_fwd(means, quats, scales, colors, opac, bg, None, W, H, 16, args_cam[0],
                                                      args_cam[1], args_cam[2], cam_model, ut, shutter, *dist, off, fl)
    bwd = ops.rasterize_to_pixels_from_world_3dgs_bwd(means, quats, scales, colors, opac, bg, None, W, H, 16, args_cam[0],
                                                      args_cam[1], args_cam[2], cam_model, ut, shutter, *dist, off, fl,
                                                      t(o["alphas"]), t(o["last_ids"]), t(v_rc), t(v_ra))
    return proj, fwd, bwd


def _check(proj, fwd, bwd, o, min_ok=0.9):
    radii = proj[0].cpu().numpy()
    vis_g, vis_o = (radii > 0).all(-1), (o["radii"] > 0).all(-1)
    assert (vis_g != vis_o).mean() < 5e-3
    both = vis_g & vis_o
    assert both.mean() > 0.2
    assert np.abs(radii[both] - o["radii"][both]).max() <= 1
    assert np.abs(np32(proj[1])[both] - o["means2d"][both]).max() < 5e-2
    np.testing.assert_allclose(np32(proj[2])[both], o["depths"][both], rtol=1e-5, atol=1e-5)
    ok = o["fragile"] == 0
    assert ok.mean() > min_ok
    assert np.abs(np32(fwd[0]) - o["renders"])[ok].max() < 1e-4
    assert np.abs(np32(fwd[1]) - o["alphas"])[ok].max() < 1e-4
    assert np.array_equal(fwd[2].cpu().numpy()[ok], o["last_ids"][ok])
    for name, g in zip(["v_means", "v_quats", "v_scales", "v_colors", "v_opacities"], bwd):
        e = rel_l2(np32(g), o[name])
        assert e < 1e-3, (name, e)


def _grads(size):
    rng = np.random.default_rng(3)
    return (rng.standard_normal((1, size, size, 3)).astype(np.float32), rng.standard_normal((1, size, size, 1)).astype(np.float32))


def test_distorted_pinhole(mods):
    ops, scenes = mods
    sc = _scene(scenes)
    radial = np.array([[0.05, -0.02, 0.003, 0.0, 0.0, 0.0]], np.float32)
    tang = np.array([[0.002, -0.001]], np.float32)
    prism = np.array([[0.001, 0.0, -0.001, 0.0]], np.float32)
    v_rc, v_ra = _grads(128)
    cam = dict(camera_model=oracle.PINHOLE, radial=radial, tangential=tang, thin_prism=prism)
    o = oracle_pipeline(sc, frag_rel=1e-3, v_render_colors=v_rc, v_render_alphas=v_ra, cam=cam)
    res = _run_gpu(ops, sc, o, ops.CameraModelType.PINHOLE, ops.ShutterType.GLOBAL, None, radial, tang, prism, v_rc, v_ra)
    _check(*res, o)


def test_fisheye(mods):
    ops, scenes = mods
    sc = _scene(scenes, f=70.0)
    radial = np.array([[0.02, -0.005, 0.001, 0.0]], np.float32)
    v_rc, v_ra = _grads(128)
    cam = dict(camera_model=oracle.FISHEYE, radial=radial)
    o = oracle_pipeline(sc, frag_rel=1e-3, v_render_colors=v_rc, v_render_alphas=v_ra, cam=cam)
    res = _run_gpu(ops, sc, o, ops.CameraModelType.FISHEYE, ops.ShutterType.GLOBAL, None, radial, None, None, v_rc, v_ra)
    _check(*res, o)


def _opaque_scene(scenes, N=900, size=128, f=90.0, seed=29):
    """A third of the Gaussians opaque (opacity 1.0, 0.9995, 0.9992) and large enough on screen (sigma ~ 10 px) that pixels next to
    their centres see opacity x exp(-s) above 0.999: the alpha clamp of Fwd.cu:239 and the gradient mask of Bwd.cu:318 — a regime none
    of the random scenes reaches (their opacities stop at 0.9) and one a trained model lives in (sigmoid(raw) -> 1)."""
    sc = _scene(scenes, N=N, size=size, seed=seed, f=f)
    g = torch.Generator().manual_seed(seed)
    n_op = N // 3
    sc["opacities"][:n_op] = torch.tensor([1.0, 0.9995, 0.9992])[torch.arange(n_op) % 3]
    sc["scales"][:n_op] = torch.rand(n_op, 3, generator=g) * 0.2 + 0.2
    perm = torch.randperm(N, generator=g)   # opaque and ordinary Gaussians interleaved in depth order AND in index order
    for k in ("means", "quats", "scales", "opacities", "sh"):
        sc[k] = sc[k][perm].contiguous()
    return sc


@pytest.mark.parametrize("bwd_kernel", ["pm", "gq"])
@pytest.mark.parametrize("camera", ["pinhole", "distorted", "fisheye", "rolling"])
def test_opaque_gaussians_alpha_clamp(mods, monkeypatch, camera, bwd_kernel):
    """alpha = min(0.999, o exp(-s)) and `clamped alpha carries no gradient` on every kernel family: the fast path's forward kernels
    (fixture) x its two backward kernels (their clamped instantiations: gq_row<true, ...>), the distorted / fisheye variants and the
    reference-order kernels (rolling shutter).  The scene is shown to exercise the clamp: the oracle's gradients with the opacities
    held just below 0.999 differ from the real ones by far more than the tolerance."""
    ops, scenes = mods
    monkeypatch.setenv("GSX_BWD", bwd_kernel)
    sc = _opaque_scene(scenes, f=70.0 if camera == "fisheye" else 90.0)
    v_rc, v_ra = _grads(128)
    kw, run = {}, dict(cam_model=ops.CameraModelType.PINHOLE, shutter=ops.ShutterType.GLOBAL, viewmats1=None, radial=None, tangential=None, thin_prism=None)
    if camera == "distorted":
        run.update(radial=np.array([[0.05, -0.02, 0.003, 0.0, 0.0, 0.0]], np.float32), tangential=np.array([[0.002, -0.001]], np.float32))
        kw = dict(camera_model=oracle.PINHOLE, radial=run["radial"], tangential=run["tangential"])
    elif camera == "fisheye":
        run.update(cam_model=ops.CameraModelType.FISHEYE, radial=np.array([[0.02, -0.005, 0.001, 0.0]], np.float32))
        kw = dict(camera_model=oracle.FISHEYE, radial=run["radial"])
    elif camera == "rolling":
        vm1 = sc["viewmat"].clone()
        vm1[0, 3], vm1[1, 3] = 0.03, -0.02
        run.update(shutter=ops.ShutterType.ROLLING_TOP_TO_BOTTOM, viewmats1=vm1[None].numpy())
        kw = dict(camera_model=oracle.PINHOLE, shutter=int(ops.ShutterType.ROLLING_TOP_TO_BOTTOM), viewmats1=vm1[None].numpy())
    o = oracle_pipeline(sc, frag_rel=1e-3, v_render_colors=v_rc, v_render_alphas=v_ra, cam=kw)
    below = dict(sc)
    below["opacities"] = sc["opacities"].clamp(max=0.9989)
    o_below = oracle_pipeline(below, v_render_colors=v_rc, v_render_alphas=v_ra, cam=kw, isect_override=(o["tile_offsets"], o["flatten_ids"]))
    assert rel_l2(o_below["v_opacities"], o["v_opacities"]) > 0.03 and rel_l2(o_below["v_means"], o["v_means"]) > 0.005   # the clamp matters here (measured: 0.05 - 0.20 / 0.007 - 0.04)
    res = _run_gpu(ops, sc, o, run["cam_model"], run["shutter"], run["viewmats1"], run["radial"], run["tangential"], run["thin_prism"], v_rc, v_ra)
    _check(*res, o, min_ok=0.85)
    clamped = (sc["opacities"] >= 0.9992).numpy() & (o["radii"] > 0).all(-1)[0]
    assert clamped.sum() > 100
    gv = np32(res[2][4]).reshape(-1)[clamped]
    assert rel_l2(gv, o["v_opacities"].reshape(-1)[clamped]) < 1e-3   # the opaque Gaussians' own opacity gradients (what the mask removes terms from)


def _wide_scene(N=3000, size=160, f=45.0, seed=21):
    """Gaussians all around the optical axis out to 100 degrees, seen by a ~200 degree fisheye (image radius / f = 1.78 rad)."""
    import math
    import gsx  # noqa: F401
    from gsx import scenes
    g = torch.Generator().manual_seed(seed)
    theta = torch.rand(N, generator=g) * math.radians(100.0)
    phi = torch.rand(N, generator=g) * 2 * math.pi
    dist = 2.0 + torch.rand(N, generator=g)
    means = torch.stack([torch.sin(theta) * torch.cos(phi), torch.sin(theta) * torch.sin(phi), torch.cos(theta)], 1) * dist[:, None]
    quats = torch.nn.functional.normalize(torch.randn(N, 4, generator=g), dim=-1)
    scales = torch.rand(N, 3, generator=g) * 0.05 + 0.01
    opac = torch.rand(N, generator=g) * 0.5 + 0.3
    sh = (torch.rand(N, 1, 3, generator=g) - 0.5) * 0.3
    return dict(means=means, quats=quats, scales=scales, opacities=opac, sh=sh, sh_degree=0, viewmat=torch.eye(4),
                K=scenes.intrinsics(f, f, size / 2.0, size / 2.0), width=size, height=size, background=torch.tensor([0.05, 0.1, 0.15]))


def test_fisheye_wide_field_of_view(mods):
    """Rays at and beyond 90 degrees (w <= 0 in the Delta-form) and Gaussians without a usable (u0, v0) chart: the tiles that list
    one of those go to the reference-order kernels, the rest to the fast path, and the frame as a whole must match the oracle."""
    ops, scenes = mods
    sc = _wide_scene()
    radial = np.array([[0.01, -0.002, 0.0, 0.0]], np.float32)
    v_rc, v_ra = _grads(160)
    cam = dict(camera_model=oracle.FISHEYE, radial=radial)
    o = oracle_pipeline(sc, frag_rel=1e-3, v_render_colors=v_rc, v_render_alphas=v_ra, cam=cam)
    vis = (o["radii"] > 0).all(-1)[0]
    cos_t = (sc["means"][:, 2] / sc["means"].norm(dim=1)).numpy()
    assert (vis & (cos_t < 0.12)).sum() > 20 and (vis & (cos_t > 0.5)).sum() > 200   # both regimes are on screen
    res = _run_gpu(ops, sc, o, ops.CameraModelType.FISHEYE, ops.ShutterType.GLOBAL, None, radial, None, None, v_rc, v_ra)
    _check(*res, o, min_ok=0.85)


@pytest.mark.parametrize("shutter", ["ROLLING_TOP_TO_BOTTOM", "ROLLING_LEFT_TO_RIGHT"])
def test_rolling_shutter(mods, shutter):
    ops, scenes = mods
    sc = _scene(scenes)
    vm1 = sc["viewmat"].clone()
    vm1[0, 3] = 0.03
    vm1[1, 3] = -0.02
    v_rc, v_ra = _grads(128)
    sh = getattr(ops.ShutterType, shutter)
    cam = dict(camera_model=oracle.PINHOLE, shutter=int(sh), viewmats1=vm1[None].numpy())
    o = oracle_pipeline(sc, frag_rel=1e-3, v_render_colors=v_rc, v_render_alphas=v_ra, cam=cam)
    res = _run_gpu(ops, sc, o, ops.CameraModelType.PINHOLE, sh, vm1[None].numpy(), None, None, None, v_rc, v_ra)
    _check(*res, o)


def test_two_cameras_in_one_call(mods):
    """C = 2: projection, intersect (camera bits in the key), offsets and blend with per-camera colours/opacities."""
    ops, scenes = mods
    sc = _scene(scenes, N=2000, size=96)
    vm = torch.stack([sc["viewmat"], sc["viewmat"].clone()])
    vm[1, 0, 3] = 0.1
    Ks = torch.stack([sc["K"], sc["K"]])
    N, W, H = 2000, 96, 96
    means, quats, scales, opac = (sc[k].numpy() for k in ("means", "quats", "scales", "opacities"))
    radii, means2d, depths, conics, _ = oracle.projection_ut(means, quats, scales, opac, vm.numpy(), Ks.numpy(), W, H)
    tw = th = 6
    tpg, ids, fl = oracle.intersect_tile(means2d, radii, depths, 2, 16, tw, th, True)
    off = oracle.intersect_offset(ids, 2, tw, th)
    rng = np.random.default_rng(5)
    colors = rng.random((2, N, 3)).astype(np.float32)
    opac2 = np.stack([opac, opac * 0.9]).astype(np.float32)
    bg = np.array([[0.1, 0.2, 0.3], [0.3, 0.2, 0.1]], np.float32)
    ren, alp, last, frag = oracle.rasterize_fwd(means, quats, scales, colors, opac2, bg, None, W, H, 16, vm.numpy(), Ks.numpy(), off, fl,
                                                frag_rel=1e-3)
    ut = ops.UnscentedTransformParameters()
    g_proj = ops.projection_ut_3dgs_fused(t(means), t(quats), t(scales), t(opac), vm.to(DEV), None, Ks.to(DEV), W, H, 0.3, 0.01, 1e4,
                                          0.0, False, ops.CameraModelType.PINHOLE, ut, ops.ShutterType.GLOBAL, None, None, None)
    assert (g_proj[0].cpu().numpy() != radii).mean() < 2e-2
    g_tpg, g_ids, g_fl = ops.intersect_tile(t(means2d), t(radii), t(depths), None, None, 2, 16, tw, th, True)
    assert np.array_equal(g_ids.cpu().numpy(), ids) and np.array_equal(g_fl.cpu().numpy(), fl)
    g_off = ops.intersect_offset(g_ids, 2, tw, th)
    assert np.array_equal(g_off.cpu().numpy(), off)
    g_ren, g_alp, g_last = ops.rasterize_to_pixels_from_world_3dgs_fwd(
        t(means), t(quats), t(scales), t(colors), t(opac2), t(bg), None, W, H, 16, vm.to(DEV), None, Ks.to(DEV),
        ops.CameraModelType.PINHOLE, ut, ops.ShutterType.GLOBAL, None, None, None, g_off, g_fl)
    ok = frag == 0
    assert np.abs(np32(g_ren) - ren)[ok].max() < 1e-4
    assert np.array_equal(g_last.cpu().numpy()[ok], last[ok])
    v_rc = rng.standard_normal((2, H, W, 3)).astype(np.float32)
    v_ra = rng.standard_normal((2, H, W, 1)).astype(np.float32)
    ref = oracle.rasterize_bwd(means, quats, scales, colors, opac2, bg, None, W, H, 16, vm.numpy(), Ks.numpy(), off, fl, alp, last, v_rc, v_ra)
    got = ops.rasterize_to_pixels_from_world_3dgs_bwd(
        t(means), t(quats), t(scales), t(colors), t(opac2), t(bg), None, W, H, 16, vm.to(DEV), None, Ks.to(DEV),
        ops.CameraModelType.PINHOLE, ut, ops.ShutterType.GLOBAL, None, None, None, g_off, g_fl, t(alp), t(last), t(v_rc), t(v_ra))
    for name, g, r in zip(["v_means", "v_quats", "v_scales", "v_colors", "v_opacities"], got, ref):
        assert rel_l2(np32(g), r) < 1e-3, name


def test_tile_masks(mods):
    ops, scenes = mods
    sc = _scene(scenes, N=1500, size=64)
    o = oracle_pipeline(sc, frag_rel=1e-3)
    masks = np.zeros((1, 4, 4), bool)
    masks[0, 1:3, :] = True
    ut = ops.UnscentedTransformParameters()
    ren, alp, last = ops.rasterize_to_pixels_from_world_3dgs_fwd(
        sc["means"].to(DEV), sc["quats"].to(DEV), sc["scales"].to(DEV), t(o["colors"]), sc["opacities"][None].to(DEV),
        sc["background"][None].to(DEV), t(masks), 64, 64, 16, sc["viewmat"][None].to(DEV), None, sc["K"][None].to(DEV),
        ops.CameraModelType.PINHOLE, ut, ops.ShutterType.GLOBAL, None, None, None, t(o["tile_offsets"]), t(o["flatten_ids"]))
    ren = np32(ren)[0]
    bg = sc["background"].numpy()
    assert np.allclose(ren[:16], bg) and np.allclose(ren[48:], bg)          # masked tiles: background (Fwd.cu:143-150)
    ok = (o["fragile"][0] == 0)[16:48]
    assert np.abs(ren[16:48] - o["renders"][0, 16:48])[ok].max() < 1e-4
