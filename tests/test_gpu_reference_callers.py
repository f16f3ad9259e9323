"""VERDICT r04 "missing" #2 / "next" #4: the reference's OWN callers on the drop-in.  src/training/rasterization/rasterizer.cpp (gs::training::rasterize)
and rasterizer_autograd.cpp (its three autograd functions) are compiled UNMODIFIED (oracle/build_ref_callers.sh) against compat/gsplat + libgsx_gsplat_backend.so
— the swap INTEGRATION.md describes — and, a second time, against the reference's gsplat/ headers and its own kernels compiled for gfx950.  The same
raw parameters and camera go through both: images within north_star's 1e-4 RGB L-inf (full frames: outside threshold-ambiguous pixels, as everywhere
else), the six parameter gradients within 1e-3 rel-L2, and the time per frame of the reference's glue on either backend is recorded next to
gsx's own fused render (what a reference user gets on day one vs what the fused glue adds)."""
import time

import pytest
import torch

from oracle import ref_callers
from tests.helpers import parity_record, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NAMES = ["means", "sh0", "shN", "scaling_raw", "rotation_raw", "opacity_raw"]


@pytest.fixture(scope="module")
def mods():
    g, r = ref_callers.load("gsx"), ref_callers.load("ref")
    if g is None or r is None:
        pytest.skip("oracle/_ref/gsplat_ref_callers_*.so not built (needs /root/reference at build time: oracle/build_ref_callers.sh)")
    return g, r


def _raw_params(sc):
    """Raw parameter leaves whose activations reproduce the scene (splat_data.cpp:267-286), in the reference's layout (sh0 [N,1,3], shN [N,K-1,3])."""
    op = sc["opacities"].clamp(1e-6, 1 - 1e-6)
    p = dict(means=sc["means"], sh0=sc["sh"][:, :1].contiguous(), shN=sc["sh"][:, 1:].contiguous(), scaling_raw=torch.log(sc["scales"]),
             rotation_raw=sc["quats"], opacity_raw=torch.logit(op).unsqueeze(-1))
    return {k: v.to(DEV).clone().requires_grad_(True) for k, v in p.items()}


def _render(mod, P, sc, v_img, v_alpha):
    vm, K = sc["viewmat"], sc["K"]
    for t in P.values():
        t.grad = None
    img, alpha, radii = mod.render(P["means"], P["sh0"], P["shN"], P["scaling_raw"], P["rotation_raw"], P["opacity_raw"], sc["sh_degree"],
                                   vm[:3, :3].contiguous(), vm[:3, 3].contiguous(), float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]),
                                   sc["width"], sc["height"], sc["background"].to(DEV), torch.empty(0), torch.empty(0), 0)
    ((img * v_img).sum() + (alpha * v_alpha).sum()).backward()
    torch.cuda.synchronize()
    return img.detach(), alpha.detach(), {k: P[k].grad.detach().clone() for k in NAMES}


def _time(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def _compare(mods, sc, tag, strict):
    gsx_mod, ref_mod = mods
    g = torch.Generator(device=DEV).manual_seed(5)
    H, W = sc["height"], sc["width"]
    v_img, v_alpha = torch.randn(3, H, W, device=DEV, generator=g), torch.randn(1, H, W, device=DEV, generator=g)
    P = _raw_params(sc)
    img_r, alp_r, grad_r = _render(ref_mod, P, sc, v_img, v_alpha)
    img_g, alp_g, grad_g = _render(gsx_mod, P, sc, v_img, v_alpha)
    err = (img_g - img_r).abs().amax(0)
    rec = parity_record("%s: the reference's gs::training::rasterize on the gsx drop-in vs on the reference's own kernels — image" % tag, pixels=int(err.numel()),
                        rgb_max_err=float(err.max()), rgb_pixels_over_1e4=int((err > 1e-4).sum()), alpha_max_err=float((alp_g - alp_r).abs().max()))
    grads = parity_record("%s: ... — parameter gradients (rel-L2)" % tag, **{k: rel_l2(grad_g[k].cpu().numpy(), grad_r[k].cpu().numpy()) for k in NAMES})
    if strict:
        assert rec["rgb_max_err"] < 1e-4 and rec["alpha_max_err"] < 1e-4, rec   # north_star: 1e-4 RGB L-inf, every pixel
    else:   # full frames: own projection + binning on both sides (cull / radius flips move whole Gaussians), threshold-ambiguous pixels
        assert rec["rgb_pixels_over_1e4"] <= 2e-3 * rec["pixels"] and rec["rgb_max_err"] < 0.06, rec
    for k in NAMES:
        assert grads[k] < 1e-3, (k, grads)   # north_star: 1e-3 gradient rel-L2
    return P, v_img, v_alpha


def _render_cam(mod, P, sc, v_img, v_alpha, radial, tangential, camera_model):
    vm, K = sc["viewmat"], sc["K"]
    for t in P.values():
        t.grad = None
    img, alpha, radii = mod.render(P["means"], P["sh0"], P["shN"], P["scaling_raw"], P["rotation_raw"], P["opacity_raw"], sc["sh_degree"],
                                   vm[:3, :3].contiguous(), vm[:3, 3].contiguous(), float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]),
                                   sc["width"], sc["height"], sc["background"].to(DEV), radial, tangential, camera_model)
    ((img * v_img).sum() + (alpha * v_alpha).sum()).backward()
    torch.cuda.synchronize()
    return img.detach(), alpha.detach(), {k: P[k].grad.detach().clone() for k in NAMES}


@pytest.mark.parametrize("camera", ["opencv_pinhole", "fisheye"])
def test_distorted_cameras_through_the_reference_callers(mods, camera):
    """gs::training::rasterize with a distorted camera (rasterizer.cpp:183-207 hands the camera's radial / tangential coefficients to every operator): a COLMAP
    OPENCV-style pinhole (k1, k2, p1, p2) and an equidistant fisheye (k1 .. k4) on the drop-in against the same call on the reference's kernels.  The pinhole
    kernels read SIX radial coefficients; the reference's glue pads a camera's to FOUR and its kernels read two floats past the tensor — the drop-in pads to six
    (csrc/ops_shim.cpp: make_cams), so the two-coefficient camera must give, bit for bit, what the same camera with six explicit coefficients gives; the reference
    side is given the six explicit ones (defined behaviour on both sides)."""
    import gsx  # noqa: F401
    from gsx import scenes
    gsx_mod, ref_mod = mods
    sc = scenes.scene_small()
    H, W = sc["height"], sc["width"]
    g = torch.Generator(device=DEV).manual_seed(9)
    v_img, v_alpha = torch.randn(3, H, W, device=DEV, generator=g), torch.randn(1, H, W, device=DEV, generator=g)
    P = _raw_params(sc)
    if camera == "opencv_pinhole":
        short, full, tang, model = torch.tensor([0.05, -0.02]), torch.tensor([0.05, -0.02, 0.0, 0.0, 0.0, 0.0]), torch.tensor([0.002, -0.001]), 0
    else:
        short = full = torch.tensor([0.02, -0.005, 0.001, 0.0])
        tang, model = torch.empty(0), 2
    img_r, alp_r, grad_r = _render_cam(ref_mod, P, sc, v_img, v_alpha, full, tang, model)
    img_g, alp_g, grad_g = _render_cam(gsx_mod, P, sc, v_img, v_alpha, short, tang, model)
    img_f, alp_f, grad_f = _render_cam(gsx_mod, P, sc, v_img, v_alpha, full, tang, model)
    assert torch.equal(img_g, img_f) and torch.equal(alp_g, alp_f)   # short coefficient vectors = zero-padded ones
    img_p, _, _ = _render_cam(gsx_mod, P, sc, v_img, v_alpha, torch.empty(0), torch.empty(0), model)
    rec = parity_record("cfg1 through a distorted camera (%s): the reference's gs::training::rasterize on the gsx drop-in vs on the reference's own kernels" % camera,
                        rgb_max_err=float((img_g - img_r).abs().max()), alpha_max_err=float((alp_g - alp_r).abs().max()),
                        distortion_moves_the_image_by=float((img_g - img_p).abs().max()), **{k: rel_l2(grad_g[k].cpu().numpy(), grad_r[k].cpu().numpy()) for k in NAMES})
    assert rec["rgb_max_err"] < 1e-4 and rec["alpha_max_err"] < 1e-4 and rec["distortion_moves_the_image_by"] > 1e-2, rec
    for k in NAMES:
        assert rec[k] < 1e-3, (k, rec)


def test_antialiased_render_through_the_reference_callers(mods):
    """gs::training::rasterize(..., antialiased = true) — the reference's `--antialiasing`: the projection is called with calc_compensations (rasterizer.cpp:181,241;
    the glue of this path then leaves the factors unused) — on the drop-in against the same call on the reference's kernels."""
    import gsx  # noqa: F401
    from gsx import scenes
    gsx_mod, ref_mod = mods
    if not hasattr(gsx_mod, "render_antialiased"):
        pytest.skip("oracle/_ref/gsplat_ref_callers_*.so predate render_antialiased (rebuild: oracle/build_ref_callers.sh)")
    sc = scenes.scene_small()
    H, W = sc["height"], sc["width"]
    g = torch.Generator(device=DEV).manual_seed(21)
    v_img, v_alpha = torch.randn(3, H, W, device=DEV, generator=g), torch.randn(1, H, W, device=DEV, generator=g)
    P = _raw_params(sc)
    vm, K = sc["viewmat"], sc["K"]

    def run(mod, aa):
        for t in P.values():
            t.grad = None
        args = (P["means"], P["sh0"], P["shN"], P["scaling_raw"], P["rotation_raw"], P["opacity_raw"], sc["sh_degree"], vm[:3, :3].contiguous(), vm[:3, 3].contiguous(),
                float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), W, H, sc["background"].to(DEV))
        img, alpha, _ = mod.render_antialiased(*args) if aa else mod.render(*args, torch.empty(0), torch.empty(0), 0)
        ((img * v_img).sum() + (alpha * v_alpha).sum()).backward()
        torch.cuda.synchronize()
        return img.detach(), alpha.detach(), {k: P[k].grad.detach().clone() for k in NAMES}

    img_r, alp_r, grad_r = run(ref_mod, True)
    img_g, alp_g, grad_g = run(gsx_mod, True)
    img_p, _, _ = run(gsx_mod, False)
    rec = parity_record("cfg1, antialiased: the reference's gs::training::rasterize on the gsx drop-in vs on the reference's own kernels", rgb_max_err=float((img_g - img_r).abs().max()),
                        alpha_max_err=float((alp_g - alp_r).abs().max()), antialiasing_moves_the_image_by=float((img_g - img_p).abs().max()),
                        **{k: rel_l2(grad_g[k].cpu().numpy(), grad_r[k].cpu().numpy()) for k in NAMES})
    # (measured: 0.0 on both backends — on the --gut path the reference computes the compensations and its world-space blend never sees them)
    assert rec["rgb_max_err"] < 1e-4 and rec["alpha_max_err"] < 1e-4, rec
    for k in NAMES:
        assert rec[k] < 1e-3, (k, rec)


def test_cfg1_reference_callers_on_the_drop_in(mods):
    """BASELINE configs[0]: 10 k Gaussians, SH degree 0, 256 x 256."""
    import gsx  # noqa: F401
    from gsx import scenes
    _compare(mods, scenes.scene_small(), "cfg1 (10k, 256x256)", strict=True)


def test_s1m_reference_callers_on_the_drop_in(mods):
    """BASELINE configs[1]: 1 M Gaussians, SH degree 3, 1920 x 1080 — and the time per frame (forward + backward) of the reference's glue on both
    backends, next to gsx's fused render of the same frame."""
    import gsx  # noqa: F401
    from gsx import rasterizer, scenes
    sc = scenes.scene_1m()
    P, v_img, v_alpha = _compare(mods, sc, "S-1M @1080p", strict=False)
    gsx_mod, ref_mod = mods
    t_ref = _time(lambda: _render(ref_mod, P, sc, v_img, v_alpha))
    t_gsx = _time(lambda: _render(gsx_mod, P, sc, v_img, v_alpha))
    model = scenes.to_splat_data(sc, DEV)
    for p in model.params():
        p.requires_grad_(True)
    cam = rasterizer.Camera(viewmat=sc["viewmat"].to(DEV), K=sc["K"].to(DEV), width=sc["width"], height=sc["height"])
    bg = sc["background"].to(DEV)

    def fused():
        out = rasterizer.rasterize_fused(cam, model, bg)
        ((out.render_hwc * v_img.permute(1, 2, 0)[None]).sum() + (out.alpha * v_alpha).sum()).backward()
    t_fused = _time(fused)
    parity_record("S-1M @1080p: ms per frame, forward + backward (no loss, no optimizer; torch sums as the loss)",
                  reference_glue_on_reference_kernels=round(t_ref, 3), reference_glue_on_gsx_drop_in=round(t_gsx, 3), gsx_rasterize_fused=round(t_fused, 3))
    assert t_gsx < t_ref   # the drop-in must at least not be slower behind the reference's own glue
