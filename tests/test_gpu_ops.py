"""GPU parity tests: every HIP operator, called through the C++ shim over the C ABI, against the CPU oracle
and the golden vectors.  Tolerances: integer / index work bit-exact; SH allclose(1e-4,1e-4) (the reference
test's own tolerance); blend forward 1e-4 RGB L-inf; blend backward 1e-3 gradient rel-L2 (BASELINE.json)."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle
from tests.helpers import np32, oracle_pipeline, rel_l2

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DEV = "cuda:0"


@pytest.fixture(scope="module")
def gsx_mod():
    import gsx
    from gsx import ops, rasterizer, scenes
    return gsx, ops, rasterizer, scenes


def t(a, dtype=None):
    x = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        x = x.to(dtype)
    return x.to(DEV)


@pytest.mark.parametrize("deg", [0, 1, 2, 3, 4])
def test_sh_fwd_bwd_vs_golden_and_oracle(gsx_mod, deg):
    _, ops, _, _ = gsx_mod
    g = np.load(os.path.join(GOLDEN, "sh_torch_impl.npz"))
    dirs, coeffs, vcol = t(g["dirs"]), t(g["coeffs"]), t(g["v_colors"])
    colors = ops.spherical_harmonics_fwd(deg, dirs, coeffs, None)
    np.testing.assert_allclose(np32(colors), g[f"colors_{deg}"], rtol=1e-4, atol=1e-4)
    v_coeffs, v_dirs = ops.spherical_harmonics_bwd(25, deg, dirs, coeffs, None, vcol, True)
    np.testing.assert_allclose(np32(v_coeffs), g[f"v_coeffs_{deg}"], rtol=1e-4, atol=1e-4)
    if deg > 0:
        np.testing.assert_allclose(np32(v_dirs), g[f"v_dirs_{deg}"], rtol=1e-4, atol=1e-4)
    else:
        assert torch.all(v_dirs == 0)
    # masks: masked rows untouched in fwd, zero in bwd
    masks = torch.arange(dirs.shape[0], device=DEV) % 3 != 0
    v_coeffs_m, v_dirs_m = ops.spherical_harmonics_bwd(25, deg, dirs, coeffs, masks, vcol, True)
    assert torch.all(v_coeffs_m[~masks] == 0) and torch.all(v_dirs_m[~masks] == 0)
    assert torch.equal(v_coeffs_m[masks], v_coeffs[masks])


@pytest.mark.parametrize("n,K,deg", [(1, 1, 0), (63, 4, 1), (257, 9, 2), (1000, 16, 3), (5000, 16, 2), (777, 25, 4)])
def test_sh_ragged_sizes_vs_oracle(gsx_mod, n, K, deg):
    _, ops, _, _ = gsx_mod
    rng = np.random.default_rng(n)
    dirs = rng.standard_normal((n, 3)).astype(np.float32)
    coeffs = rng.standard_normal((n, K, 3)).astype(np.float32)
    vcol = rng.standard_normal((n, 3)).astype(np.float32)
    masks = rng.random(n) > 0.3
    col = ops.spherical_harmonics_fwd(deg, t(dirs), t(coeffs), t(masks))
    ref = oracle.sh_fwd(deg, dirs, coeffs, masks)
    np.testing.assert_allclose(np32(col)[masks], ref[masks], rtol=1e-4, atol=1e-4)
    v_coeffs, v_dirs = ops.spherical_harmonics_bwd(K, deg, t(dirs), t(coeffs), t(masks), t(vcol), True)
    r_vc, r_vd = oracle.sh_bwd(deg, dirs, coeffs, masks, vcol, True)
    np.testing.assert_allclose(np32(v_coeffs), r_vc, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(np32(v_dirs), r_vd, rtol=1e-4, atol=2e-4)


def test_sh_empty(gsx_mod):
    _, ops, _, _ = gsx_mod
    out = ops.spherical_harmonics_fwd(3, torch.zeros(0, 3, device=DEV), torch.zeros(0, 16, 3, device=DEV), None)
    assert out.shape == (0, 3)


@pytest.mark.parametrize("name", ["isect_torch_impl.npz", "isect_torch_impl_256.npz"])
def test_intersect_exact_vs_golden(gsx_mod, name):
    _, ops, _, _ = gsx_mod
    g = np.load(os.path.join(GOLDEN, name))
    C = g["depths"].shape[0]
    T, tw, th = int(g["tile_size"]), int(g["tile_width"]), int(g["tile_height"])
    tpg, ids, fl = ops.intersect_tile(t(g["means2d"]), t(g["radii"]), t(g["depths"]), None, None, C, T, tw, th, True)
    assert np.array_equal(tpg.cpu().numpy(), g["tiles_per_gauss"])
    assert np.array_equal(ids.cpu().numpy(), g["isect_ids"])
    o_tpg, o_ids, o_fl = oracle.intersect_tile(g["means2d"], g["radii"], g["depths"], C, T, tw, th, True)
    assert np.array_equal(fl.cpu().numpy(), o_fl)  # stable order, bit-exact vs the oracle
    off = ops.intersect_offset(ids, C, tw, th)
    assert np.array_equal(off.cpu().numpy(), oracle.intersect_offset(o_ids, C, tw, th))
    # unsorted variant
    _, ids_u, fl_u = ops.intersect_tile(t(g["means2d"]), t(g["radii"]), t(g["depths"]), None, None, C, T, tw, th, False)
    _, o_ids_u, o_fl_u = oracle.intersect_tile(g["means2d"], g["radii"], g["depths"], C, T, tw, th, False)
    assert np.array_equal(ids_u.cpu().numpy(), o_ids_u) and np.array_equal(fl_u.cpu().numpy(), o_fl_u)


def test_intersect_empty_and_all_culled(gsx_mod):
    _, ops, _, _ = gsx_mod
    m = torch.zeros(1, 0, 2, device=DEV)
    tpg, ids, fl = ops.intersect_tile(m, torch.zeros(1, 0, 2, dtype=torch.int32, device=DEV), torch.zeros(1, 0, device=DEV),
                                      None, None, 1, 16, 4, 4, True)
    assert ids.numel() == 0 and fl.numel() == 0
    off = ops.intersect_offset(ids, 1, 4, 4)
    assert torch.all(off == 0)
    m = torch.rand(1, 100, 2, device=DEV) * 64
    tpg, ids, fl = ops.intersect_tile(m, torch.zeros(1, 100, 2, dtype=torch.int32, device=DEV), torch.rand(1, 100, device=DEV),
                                      None, None, 1, 16, 4, 4, True)
    assert ids.numel() == 0 and torch.all(tpg == 0)


def _scene(scenes, N=3000, size=128, seed=3, deg=0):
    sc = scenes.scene_small(seed=seed, N=N)
    sc["width"] = sc["height"] = size
    sc["K"] = scenes.intrinsics(100.0, 100.0, size / 2.0, size / 2.0)
    sc["background"] = torch.tensor([0.1, 0.2, 0.3])
    if deg > 0:
        g = torch.Generator().manual_seed(seed)
        sc["sh"] = (torch.rand(N, (deg + 1) ** 2, 3, generator=g) - 0.5) * 0.3
        sc["sh_degree"] = deg
    return sc


def _project_gpu(ops, sc):
    return ops.projection_ut_3dgs_fused(sc["means"].to(DEV), sc["quats"].to(DEV), sc["scales"].to(DEV),
                                        sc["opacities"].to(DEV), sc["viewmat"][None].to(DEV), None, sc["K"][None].to(DEV),
                                        sc["width"], sc["height"], 0.3, 0.01, 1e4, 0.0, False, ops.CameraModelType.PINHOLE,
                                        ops.UnscentedTransformParameters(), ops.ShutterType.GLOBAL, None, None, None)


def test_projection_vs_oracle(gsx_mod):
    _, ops, _, scenes = gsx_mod
    sc = _scene(scenes, N=20000, size=256)
    o = oracle_pipeline(sc)
    radii, means2d, depths, conics, comp = _project_gpu(ops, sc)
    radii, means2d, depths, conics = radii.cpu().numpy(), np32(means2d), np32(depths), np32(conics)
    vis_g, vis_o = (radii > 0).all(-1), (o["radii"] > 0).all(-1)
    assert (vis_g != vis_o).mean() < 2e-3            # cull flips only at thresholds
    both = vis_g & vis_o
    assert both.mean() > 0.5
    assert np.abs(radii[both] - o["radii"][both]).max() <= 1
    assert (radii[both] != o["radii"][both]).mean() < 2e-2
    assert np.abs(means2d[both] - o["means2d"][both]).max() < 5e-2   # UT cancellation noise, SURVEY §7
    np.testing.assert_allclose(depths[both], o["depths"][both], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(conics[both], o["conics"][both], rtol=2e-2, atol=1e-4)


def _blend_inputs(sc, o):
    return dict(means=sc["means"].to(DEV), quats=sc["quats"].to(DEV), scales=sc["scales"].to(DEV),
                colors=t(o["colors"]), opac=sc["opacities"][None].to(DEV), bg=sc["background"][None].to(DEV),
                viewmat=sc["viewmat"][None].to(DEV), K=sc["K"][None].to(DEV), off=t(o["tile_offsets"]),
                fl=t(o["flatten_ids"]))


def _fwd(ops, sc, b, bg=True):
    return ops.rasterize_to_pixels_from_world_3dgs_fwd(
        b["means"], b["quats"], b["scales"], b["colors"], b["opac"], b["bg"] if bg else None, None, sc["width"],
        sc["height"], 16, b["viewmat"], None, b["K"], ops.CameraModelType.PINHOLE, ops.UnscentedTransformParameters(),
        ops.ShutterType.GLOBAL, None, None, None, b["off"], b["fl"])


@pytest.fixture(params=["fast", "fast-quad", "fast-pair", "generic"])
def raster_path(request):
    """Both kernel families: the MI355X fast path — with each of its three forward kernels forced (one list per 8x8 quadrant / four lists per
    wave / two pixels per lane and eight lists per wave; the launcher picks by footprint size and grid) — and the reference-order generic path."""
    old = {k: os.environ.get(k) for k in ("GSX_RASTER_PATH", "GSX_FWD")}
    os.environ["GSX_RASTER_PATH"] = "generic" if request.param == "generic" else "fast"
    os.environ["GSX_FWD"] = {"fast-quad": "quad", "fast-pair": "pair"}.get(request.param, "wave")
    yield "fast" if request.param.startswith("fast") else "generic"
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


@pytest.mark.parametrize("N,size,seed", [(3000, 128, 3), (10000, 256, 42), (500, 100, 9)])
def test_blend_forward_vs_oracle(gsx_mod, raster_path, N, size, seed):
    _, ops, _, scenes = gsx_mod
    sc = _scene(scenes, N=N, size=size, seed=seed)   # size 100: ragged last tiles
    o = oracle_pipeline(sc, frag_rel=1e-3)
    b = _blend_inputs(sc, o)
    renders, alphas, last_ids = _fwd(ops, sc, b)
    ok = o["fragile"] == 0
    assert ok.mean() > 0.97
    err = np.abs(np32(renders) - o["renders"])
    assert err[ok].max() < 1e-4, err[ok].max()
    assert np.abs(np32(alphas) - o["alphas"])[ok].max() < 1e-4
    assert np.array_equal(last_ids.cpu().numpy()[ok], o["last_ids"][ok])
    # fragile pixels may flip one alpha>=1/255 decision: bounded by one Gaussian's contribution
    assert err.max() < 2.0 / 255.0 + 1e-3


def test_blend_forward_s1m_crop_accuracy(gsx_mod, raster_path):
    """A 1/16 crop of S-1M (same focal length, density, scale range, depth range): small far Gaussians, where the
    reference's fp32 cross product cancels worst.  Parity bar 1e-4 on non-fragile pixels; and the fast path must be
    at least as close to the float64 evaluation as the reference-order fp32 oracle is."""
    _, ops, _, scenes = gsx_mod
    sc = scenes.scene_frustum(62_500, 480, 270, 1000.0, (2.0, 10.0), sh_degree=0, seed=42)
    o32 = oracle_pipeline(sc, np.float32, frag_rel=2e-3)
    o64 = oracle_pipeline(sc, np.float64, isect_override=(o32["tile_offsets"], o32["flatten_ids"]), colors_override=o32["colors"])
    b = _blend_inputs(sc, o32)
    renders, alphas, last_ids = _fwd(ops, sc, b)
    ok = o32["fragile"] == 0
    assert ok.mean() > 0.9
    e_gpu32 = np.abs(np32(renders) - o32["renders"])[ok]
    e_gpu64 = np.abs(np32(renders).astype(np.float64) - o64["renders"])[ok]
    e_3264 = np.abs(o32["renders"].astype(np.float64) - o64["renders"])[ok]
    print("path=%s  max|gpu-o32|=%.2e  max|gpu-o64|=%.2e  max|o32-o64|=%.2e  mean: %.2e %.2e %.2e" % (
        raster_path, e_gpu32.max(), e_gpu64.max(), e_3264.max(), e_gpu32.mean(), e_gpu64.mean(), e_3264.mean()))
    assert e_gpu32.max() < 1e-4
    assert e_gpu64.max() < 1e-4
    if raster_path == "fast":
        assert e_gpu64.mean() <= 1.5 * e_3264.mean() + 1e-8


def test_blend_backward_vs_oracle(gsx_mod, raster_path):
    _, ops, _, scenes = gsx_mod
    sc = _scene(scenes, N=3000, size=128, seed=3)
    rng = np.random.default_rng(0)
    v_rc = rng.standard_normal((1, 128, 128, 3)).astype(np.float32)
    v_ra = rng.standard_normal((1, 128, 128, 1)).astype(np.float32)
    o = oracle_pipeline(sc, v_render_colors=v_rc, v_render_alphas=v_ra)
    b = _blend_inputs(sc, o)
    # feed the oracle's forward state so both backward passes walk exactly the same Gaussians
    grads = ops.rasterize_to_pixels_from_world_3dgs_bwd(
        b["means"], b["quats"], b["scales"], b["colors"], b["opac"], b["bg"], None, 128, 128, 16, b["viewmat"], None, b["K"],
        ops.CameraModelType.PINHOLE, ops.UnscentedTransformParameters(), ops.ShutterType.GLOBAL, None, None, None, b["off"],
        b["fl"], t(o["alphas"]), t(o["last_ids"]), t(v_rc), t(v_ra))
    for name, g in zip(["v_means", "v_quats", "v_scales", "v_colors", "v_opacities"], grads):
        e = rel_l2(np32(g), o[name])
        assert e < 1e-3, (name, e)


@pytest.mark.parametrize("variant", ["pm", "gq"])
def test_blend_backward_variants_vs_oracle(gsx_mod, variant):
    """The two backward kernels of the fast path — pixel-major and Gaussian-major (the launcher picks by footprint size) — each forced
    on an S-1M-like scene (small footprints, ragged tiles) and compared with the oracle."""
    _, ops, _, scenes = gsx_mod
    sc = scenes.scene_frustum(20_000, 320, 192, 400.0, (2.0, 10.0), sh_degree=0, seed=7)
    H, W = sc["height"], sc["width"]
    rng = np.random.default_rng(1)
    v_rc = rng.standard_normal((1, H, W, 3)).astype(np.float32)
    v_ra = rng.standard_normal((1, H, W, 1)).astype(np.float32)
    o = oracle_pipeline(sc, v_render_colors=v_rc, v_render_alphas=v_ra)
    b = _blend_inputs(sc, o)
    old = os.environ.get("GSX_BWD")
    os.environ["GSX_BWD"] = variant
    try:
        grads = ops.rasterize_to_pixels_from_world_3dgs_bwd(
            b["means"], b["quats"], b["scales"], b["colors"], b["opac"], b["bg"], None, W, H, 16, b["viewmat"], None, b["K"],
            ops.CameraModelType.PINHOLE, ops.UnscentedTransformParameters(), ops.ShutterType.GLOBAL, None, None, None, b["off"],
            b["fl"], t(o["alphas"]), t(o["last_ids"]), t(v_rc), t(v_ra))
    finally:
        if old is None:
            os.environ.pop("GSX_BWD", None)
        else:
            os.environ["GSX_BWD"] = old
    for name, g in zip(["v_means", "v_quats", "v_scales", "v_colors", "v_opacities"], grads):
        e = rel_l2(np32(g), o[name])
        assert e < 1e-3, (variant, name, e)


@pytest.mark.parametrize("variant", ["pm", "gq"])
def test_blend_backward_record_chains(gsx_mod, variant):
    """The backward chains its moment records per Gaussian — one chain on frames of small footprints, four (by tile parity) on frames of large
    ones (gsx_raster_common.hpp: tile_chain).  Both forced on a scene of LARGE footprints (chains of tens of records), both kernels:
    the oracle's gradients either way, the same sums up to the order of the additions, and a second backward on the same forward
    workspace (the gather must have put every head back to -1) repeats the first."""
    _, ops, _, scenes = gsx_mod
    sc = scenes.scene_frustum(1_500, 256, 192, 300.0, (1.5, 6.0), sh_degree=0, seed=11)
    sc["scales"] = sc["scales"] * 20.0   # footprints of many tiles
    H, W = sc["height"], sc["width"]
    rng = np.random.default_rng(2)
    v_rc = rng.standard_normal((1, H, W, 3)).astype(np.float32)
    v_ra = rng.standard_normal((1, H, W, 1)).astype(np.float32)
    o = oracle_pipeline(sc, v_render_colors=v_rc, v_render_alphas=v_ra)
    b = _blend_inputs(sc, o)
    assert b["fl"].shape[0] > 12 * 1_500   # more than 12 tiles per Gaussian on average
    args = (b["means"], b["quats"], b["scales"], b["colors"], b["opac"], b["bg"], None, W, H, 16, b["viewmat"], None, b["K"],
            ops.CameraModelType.PINHOLE, ops.UnscentedTransformParameters(), ops.ShutterType.GLOBAL, None, None, None, b["off"], b["fl"])
    fwd = ops.rasterize_to_pixels_from_world_3dgs_fwd(*args, keep_ws=True)
    ws = fwd[3]
    old = {k: os.environ.get(k) for k in ("GSX_BWD", "GSX_BWD_CHAINS")}
    res = {}
    try:
        os.environ["GSX_BWD"] = variant
        for chains in ("1", "4", "4"):   # (the third run: the heads the second one consumed are empty again)
            os.environ["GSX_BWD_CHAINS"] = chains
            g = ops.rasterize_to_pixels_from_world_3dgs_bwd(*args, fwd[1], fwd[2], t(v_rc), t(v_ra), fwd_ws=ws)
            res.setdefault(chains, []).append([np32(x) for x in g])
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    names = ["v_means", "v_quats", "v_scales", "v_colors", "v_opacities"]
    for chains, runs in res.items():
        for name, g in zip(names, runs[0]):
            assert rel_l2(g, o[name]) < 1e-3, (variant, chains, name, rel_l2(g, o[name]))
    for name, g1, g4, g4b in zip(names, res["1"][0], res["4"][0], res["4"][1]):
        assert rel_l2(g4, g1) < 1e-5 and rel_l2(g4b, g4) < 1e-5, (variant, name, rel_l2(g4, g1), rel_l2(g4b, g4))


def test_rasterize_autograd_end_to_end(gsx_mod):
    """gs::training::rasterize mirror: image parity with the oracle pipeline and gradients that flow to every
    raw parameter through the activations."""
    _, ops, rasterizer, scenes = gsx_mod
    sc = _scene(scenes, N=4000, size=128, seed=21, deg=3)
    model = scenes.to_splat_data(sc, DEV)
    for p in model.params():
        p.requires_grad_(True)
    cam = rasterizer.Camera(viewmat=sc["viewmat"].to(DEV), K=sc["K"].to(DEV), width=128, height=128)
    out = rasterizer.rasterize(cam, model, sc["background"].to(DEV))
    o = oracle_pipeline(sc, frag_rel=1e-3)
    img = out.image.permute(1, 2, 0).detach().cpu().numpy()
    ref = np.clip(o["renders"][0], 0, 1)
    diff = np.abs(img - ref)
    # binning may differ by +-1 px radii for a handful of Gaussians: allow a few outlier pixels
    assert np.quantile(diff, 0.999) < 1e-4 and diff.max() < 2e-2, (np.quantile(diff, 0.999), diff.max())
    loss = (out.image * torch.linspace(0, 1, 128, device=DEV)).sum() + out.alpha.sum()
    loss.backward()
    for p in model.params():
        assert p.grad is not None and torch.isfinite(p.grad).all() and p.grad.abs().sum() > 0


def test_error_behaviour(gsx_mod):
    _, ops, _, _ = gsx_mod
    with pytest.raises(RuntimeError):  # non-contiguous input (CHECK_INPUT upstream)
        ops.spherical_harmonics_fwd(0, torch.zeros(3, 8, device=DEV).t(), torch.zeros(8, 1, 3, device=DEV), None)
    with pytest.raises(RuntimeError):  # degree needs (deg+1)^2 <= K
        ops.spherical_harmonics_fwd(3, torch.zeros(8, 3, device=DEV), torch.zeros(8, 4, 3, device=DEV), None)


def test_blend_with_no_intersections(gsx_mod, raster_path):
    """All Gaussians culled: n_isects == 0 -> background image, zero alpha, zero gradients (Bwd.cu:434-437)."""
    _, ops, _, scenes = gsx_mod
    sc = _scene(scenes, N=64, size=48, seed=1)
    N = 64
    off = torch.zeros(1, 3, 3, dtype=torch.int32, device=DEV)
    fl = torch.zeros(0, dtype=torch.int32, device=DEV)
    colors = torch.rand(1, N, 3, device=DEV)
    ut = ops.UnscentedTransformParameters()
    args = (sc["means"].to(DEV), sc["quats"].to(DEV), sc["scales"].to(DEV), colors, sc["opacities"][None].to(DEV),
            sc["background"][None].to(DEV), None, 48, 48, 16, sc["viewmat"][None].to(DEV), None, sc["K"][None].to(DEV),
            ops.CameraModelType.PINHOLE, ut, ops.ShutterType.GLOBAL, None, None, None, off, fl)
    ren, alp, last = ops.rasterize_to_pixels_from_world_3dgs_fwd(*args)
    assert torch.allclose(ren, sc["background"].to(DEV).expand(1, 48, 48, 3)) and torch.all(alp == 0) and torch.all(last == 0)
    grads = ops.rasterize_to_pixels_from_world_3dgs_bwd(*args, alp, last, torch.randn(1, 48, 48, 3, device=DEV),
                                                        torch.randn(1, 48, 48, 1, device=DEV))
    assert all(float(g.abs().max()) == 0.0 for g in grads)


@pytest.mark.gpu
@pytest.mark.parametrize("C,N,W,H,rmax,nq", [(1, 5000, 256, 256, 20, 0), (2, 3000, 200, 120, 40, 16), (1, 30000, 64, 64, 30, 8),
                                             (3, 1, 33, 17, 5, 0), (1, 100, 1920, 1080, 300, 4),
                                             (1, 20000, 64, 64, 30, 8), (1, 50000, 64, 64, 30, 0), (2, 70000, 64, 48, 30, 0),
                                             (1, 700000, 64, 64, 30, 0), (1, 150000, 64, 64, 30, 4)])
@pytest.mark.parametrize("fill", ["keys", "ranked"])
def test_intersect_tile_binned_equals_sorted_path(C, N, W, H, rmax, nq, fill, monkeypatch):
    """The binned pipeline (LDS histograms + per-tile LDS sort, incl. the > 4096-key merge path: case 3) returns bit for bit
    what intersect_tile(sort=True) + intersect_offset return; quantised depths (nq levels) exercise the flatten-index tie break.
    fill = "keys": 64-bit (depth, index) keys, LDS merge sorts, chunk + merge passes for giant segments; "ranked": frame-wide depth
    ranks as 4-byte keys, bitmap sort for tiles above 4096 keys (the shim picks one by the previous frame's statistics)."""
    import gsx  # noqa: F401
    from gsx import ops
    monkeypatch.setenv("GSX_INTERSECT_FILL", fill)
    ops.shim_ranked_calls(True)
    g = torch.Generator().manual_seed(C * 1000 + N)
    means2d = torch.rand(C, N, 2, generator=g) * torch.tensor([W * 1.2, H * 1.2]) - torch.tensor([W * 0.1, H * 0.1])
    radii = torch.randint(0, rmax, (C, N, 2), generator=g, dtype=torch.int32)      # zeros = culled
    depths = torch.rand(C, N, generator=g) * 10 + 0.1
    if nq:
        depths = torch.round(depths * nq / 10) * 10 / nq + 0.1
    means2d, radii, depths = means2d.cuda(), radii.cuda(), depths.cuda()
    tw, th = (W + 15) // 16, (H + 15) // 16
    tpg, ids, flat = ops.intersect_tile_device_sort(means2d, radii, depths, C, 16, tw, th, True)
    off = ops.intersect_offset(ids, C, tw, th)
    tpg2, ids2, flat2, off2 = ops.intersect_tile_binned(means2d, radii, depths, C, 16, tw, th, True)
    assert torch.equal(tpg, tpg2) and torch.equal(off, off2)
    assert torch.equal(ids, ids2) and torch.equal(flat, flat2)
    _, ids3, flat3, off3 = ops.intersect_tile_binned(means2d, radii, depths, C, 16, tw, th, False)
    assert ids3.numel() == 0 and torch.equal(flat, flat3) and torch.equal(off, off3)
    # (segments up to 1024 keys: the call above sorts them with the register bitonic network, the one with isect_ids with the LDS merge
    #  sort; forced here once more without isect_ids)
    monkeypatch.setenv("GSX_WAVE_SORT", "merge")
    _, _, flat4, off4 = ops.intersect_tile_binned(means2d, radii, depths, C, 16, tw, th, False)
    monkeypatch.delenv("GSX_WAVE_SORT")
    assert torch.equal(flat, flat4) and torch.equal(off, off4)
    seg = torch.cat([off.flatten(), torch.tensor([flat.numel()], device=off.device, dtype=off.dtype)])
    seg = seg[1:] - seg[:-1]
    assert ops.shim_ranked_calls(True) == (3 if fill == "ranked" else 0)
    if N == 30000:
        assert int(seg.max()) > 4096                                         # beyond the 4096-key block sort
    if N in (20000, 50000):
        assert bool(((seg > 4096) & (seg <= 16384)).any())                   # the heavy-tile kernel (one 1024-thread block, 132 KB of LDS) ran
    if N == 70000:
        assert int(seg.max()) > 16384                                        # giant segments: LDS-sorted chunks + merge-path passes
    if N == 700000:
        assert int(seg.max()) > 8 * 16384 and int(seg.min()) > 16384         # four merge passes, run counts that are not powers of two
    if N == 150000:
        assert int(seg.max()) > 2 * 16384                                    # ties in depth across chunk borders


@pytest.mark.gpu
def test_intersect_tile_binned_optimistic_capacity_overflow():
    """The fill is launched into buffers sized from the previous total for the same problem shape; when the guess is too small
    (second call: 3x the radii) it is repeated with the exact size; when it is too large (third call) the outputs are narrowed."""
    import gsx  # noqa: F401
    from gsx import ops
    C, N, W, H = 1, 4000, 320, 240
    g = torch.Generator().manual_seed(7)
    means2d = (torch.rand(C, N, 2, generator=g) * torch.tensor([W, H])).cuda()
    depths = (torch.rand(C, N, generator=g) * 5 + 0.2).cuda()
    base = torch.randint(1, 8, (C, N, 2), generator=g, dtype=torch.int32)
    tw, th = (W + 15) // 16, (H + 15) // 16
    sizes = []
    for scale in (1, 3, 1):
        radii = (base * scale).cuda()
        tpg, ids, flat = ops.intersect_tile_device_sort(means2d, radii, depths, C, 16, tw, th, True)
        off = ops.intersect_offset(ids, C, tw, th)
        tpg2, ids2, flat2, off2 = ops.intersect_tile_binned(means2d, radii, depths, C, 16, tw, th, True)
        assert torch.equal(ids, ids2) and torch.equal(flat, flat2) and torch.equal(off, off2) and torch.equal(tpg, tpg2)
        sizes.append(flat.numel())
    assert sizes[1] > 1.3 * sizes[0]


@pytest.mark.gpu
def test_intersect_tile_binned_segment_bound_outgrown():
    """The optimistic fill launches as many merge passes as the largest segment of the previous call needs.  Second call: the tiles
    grow past 16384 keys (no passes were launched: the fill is repeated); third call: small again, the passes find nothing to do."""
    import gsx  # noqa: F401
    from gsx import ops
    C, N, W, H = 1, 60000, 64, 64
    g = torch.Generator().manual_seed(11)
    means2d = (torch.rand(C, N, 2, generator=g) * torch.tensor([W, H])).cuda()
    depths = (torch.rand(C, N, generator=g) * 5 + 0.2).cuda()
    base = torch.randint(1, 8, (C, N, 2), generator=g, dtype=torch.int32)
    tw, th = (W + 15) // 16, (H + 15) // 16
    largest = []
    for scale in (1, 3, 5, 1):
        radii = (base * scale).cuda()
        tpg, ids, flat = ops.intersect_tile_device_sort(means2d, radii, depths, C, 16, tw, th, True)
        off = ops.intersect_offset(ids, C, tw, th)
        tpg2, ids2, flat2, off2 = ops.intersect_tile_binned(means2d, radii, depths, C, 16, tw, th, True)
        assert torch.equal(ids, ids2) and torch.equal(flat, flat2) and torch.equal(off, off2) and torch.equal(tpg, tpg2)
        seg = torch.cat([off.flatten(), torch.tensor([flat.numel()], device=off.device, dtype=off.dtype)])
        largest.append(int((seg[1:] - seg[:-1]).max()))
    assert largest[0] <= 16384 < largest[1] and largest[2] > 1.3 * largest[1] and largest[3] == largest[0]


def test_shim_caches_behind_the_plain_operator_calls(gsx_mod):
    """Round 6 (csrc/ops_shim.cpp: OffsetsCache, PackCache): the plain Ops.h calls of a reference build — intersect_tile then intersect_offset on its
    isect_ids, blend forward then blend backward on the tensors the forward saved — reuse what the first call of each pair already produced
    (the binned pipeline's offsets, the packed records).  Same results as the uncached paths: a clone of the key tensor (another data pointer),
    an in-place change of an input between forward and backward (a version bump) or a forward on other tensors in between take the uncached
    path, which is the yardstick here."""
    _, ops, _, scenes = gsx_mod
    sc = _scene(scenes, N=3000, size=128, seed=3)
    rng = np.random.default_rng(0)
    v_rc, v_ra = t(rng.standard_normal((1, 128, 128, 3)).astype(np.float32)), t(rng.standard_normal((1, 128, 128, 1)).astype(np.float32))
    o = oracle_pipeline(sc)
    b = _blend_inputs(sc, o)
    ut = ops.UnscentedTransformParameters()
    # ---- (a) offsets
    P = ops.projection_ut_3dgs_fused(b["means"], b["quats"], b["scales"], b["opac"][0].contiguous(), b["viewmat"], None, b["K"], 128, 128, 0.3, 0.01, 1e4, 0.0,
                                     False, ops.CameraModelType.PINHOLE, ut, ops.ShutterType.GLOBAL, None, None, None)
    radii, means2d, depths = P[0], P[1], P[2]
    _, ids, fl = ops.intersect_tile(means2d, radii, depths, None, None, 1, 16, 8, 8, True)
    off_cached = ops.intersect_offset(ids, 1, 8, 8)            # answered from the pipeline's by-product
    off_again = ops.intersect_offset(ids, 1, 8, 8)             # the cache entry was handed out: the kernel
    off_clone = ops.intersect_offset(ids.clone(), 1, 8, 8)     # another tensor: the kernel
    assert off_cached.shape == (1, 8, 8) and torch.equal(off_cached, off_again) and torch.equal(off_cached, off_clone)
    assert int(fl.numel()) > 3000 and int(off_cached.max()) > 0
    # ---- (b) packed records
    args = lambda B: (B["means"], B["quats"], B["scales"], B["colors"], B["opac"], B["bg"], None, 128, 128, 16, B["viewmat"], None, B["K"],  # noqa: E731
                      ops.CameraModelType.PINHOLE, ut, ops.ShutterType.GLOBAL, None, None, None, B["off"], B["fl"])
    F = ops.rasterize_to_pixels_from_world_3dgs_fwd(*args(b))
    g_hit = ops.rasterize_to_pixels_from_world_3dgs_bwd(*args(b), F[1], F[2], v_rc, v_ra)                    # records from the forward
    b2 = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    g_miss = ops.rasterize_to_pixels_from_world_3dgs_bwd(*args(b2), F[1], F[2], v_rc, v_ra)                  # other tensors: packs again
    for x, y, n in zip(g_hit, g_miss, ["v_means", "v_quats", "v_scales", "v_colors", "v_opacities"]):
        assert rel_l2(np32(x), np32(y)) < 2e-6, n
    # an input changed in place between forward and backward (version bump): the backward must see the NEW values, not the forward's records
    F = ops.rasterize_to_pixels_from_world_3dgs_fwd(*args(b))
    b["means"].add_(0.003)
    g_new = ops.rasterize_to_pixels_from_world_3dgs_bwd(*args(b), F[1], F[2], v_rc, v_ra)
    b3 = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    g_ref = ops.rasterize_to_pixels_from_world_3dgs_bwd(*args(b3), F[1], F[2], v_rc, v_ra)
    assert rel_l2(np32(g_new[0]), np32(g_ref[0])) < 2e-6 and rel_l2(np32(g_new[0]), np32(g_hit[0])) > 1e-3
