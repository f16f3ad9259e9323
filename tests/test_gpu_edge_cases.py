"""Edge cases of the fused render path: image sides that are not multiples of the tile, tile lists long enough to leave the
wave-level sort (> 1024 keys) and the single-pass block sort (> 4096 keys), saturated pixels (the forward's early exit, the
backward's last_ids cut-off), an empty model — each against the CPU oracle or the reference-style chain."""
import numpy as np
import pytest
import torch

from tests.helpers import oracle_pipeline, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _dense_scene(N, W, H, opacity, seed=0):
    import gsx  # noqa: F401
    from gsx import scenes
    g = torch.Generator().manual_seed(seed)
    sc = scenes.scene_small(seed=seed, N=N)
    sc["means"] = torch.cat([(torch.rand(N, 2, generator=g) - 0.5) * 0.6, 2.0 + torch.rand(N, 1, generator=g) * 2.0], 1)
    sc["scales"] = torch.rand(N, 3, generator=g) * 0.08 + 0.05          # every Gaussian covers many tiles
    sc["opacities"] = torch.full((N,), opacity)
    sc["width"], sc["height"] = W, H
    sc["K"] = scenes.intrinsics(60.0, 60.0, W / 2.0, H / 2.0)
    return sc


@pytest.mark.parametrize("N,opacity", [(1500, 0.02), (6000, 0.01), (3000, 0.6)])
def test_dense_tiles_odd_image_size_vs_oracle(N, opacity):
    """40 x 24 px (3 x 2 tiles, ragged right / bottom edge).  1500 / 6000 Gaussians per tile exercise the block sort and the
    chunk-merge sort; opacity 0.6 saturates every pixel after a few dozen Gaussians (early exit, last_ids)."""
    import gsx  # noqa: F401
    from gsx import rasterizer, scenes
    W, H = 40, 24
    sc = _dense_scene(N, W, H, opacity, seed=N)
    model = scenes.to_splat_data(sc, DEV)
    for p in model.params():
        p.requires_grad_(True)
    cam = rasterizer.Camera(viewmat=sc["viewmat"].to(DEV), K=sc["K"].to(DEV), width=W, height=H)
    out = rasterizer.rasterize_fused(cam, model, sc["background"].to(DEV))
    per_tile = out.aux["isect_offsets"].flatten().diff()
    assert int(per_tile.max()) > (4096 if N >= 6000 else 1024)
    g = torch.Generator().manual_seed(5)
    v_rc = torch.rand(1, H, W, 3, generator=g).numpy().astype(np.float32)
    v_ra = torch.rand(1, H, W, 1, generator=g).numpy().astype(np.float32)
    (out.render_hwc * torch.from_numpy(v_rc).to(DEV)).sum().backward(retain_graph=True)
    o = oracle_pipeline(sc, frag_rel=1e-3, v_render_colors=v_rc, v_render_alphas=np.zeros_like(v_ra))
    assert out.n_isects == o["flatten_ids"].shape[0]
    assert np.array_equal(out.aux["flatten_ids"].cpu().numpy(), o["flatten_ids"])
    ok = o["fragile"][0] == 0
    assert ok.mean() > 0.8
    diff = np.abs(out.render_hwc.detach().cpu().numpy()[0] - o["renders"][0])
    assert diff[ok].max() < 1e-4
    if opacity > 0.5:
        assert float(out.alpha.detach()[0, H // 2, W // 2]) > 0.999         # saturated where the cloud is dense
    # the raw-parameter gradients of the fused path against the chain rule applied to the oracle's blend gradients
    m = model.means.grad.cpu().numpy()
    assert rel_l2(m, o["v_means"]) < 2e-3


def test_empty_model_renders_background():
    import gsx  # noqa: F401
    from gsx import rasterizer, scenes
    sc = scenes.scene_small(seed=1, N=4)
    sc["means"][:, 2] = -5.0                                               # everything behind the camera
    model = scenes.to_splat_data(sc, DEV)
    for p in model.params():
        p.requires_grad_(True)
    cam = rasterizer.Camera(viewmat=sc["viewmat"].to(DEV), K=sc["K"].to(DEV), width=64, height=48)
    bg = torch.tensor([0.2, 0.4, 0.6], device=DEV)
    out = rasterizer.rasterize_fused(cam, model, bg)
    assert out.n_isects == 0
    assert torch.allclose(out.image, bg.view(3, 1, 1).expand(3, 48, 64)) and float(out.alpha.detach().abs().max()) == 0.0
    out.image.sum().backward()
    assert all(float(p.grad.abs().max()) == 0.0 for p in model.params())


def test_collapsed_scales_give_finite_gradients():
    """a Gaussian whose footprint basis degenerates in fp32 (two scales seven orders of magnitude below the third: the cofactor columns
    B0, B1 of the Delta-form come out parallel, the orthonormal pair (q0, q1) of gsx_record.hpp has no second direction) must not turn into a
    NaN gradient — one NaN poisons the Gaussian's Adam state for good.  Every gradient of every Gaussian finite, on the fused path (Gaussian-major
    backward + gather with the activation epilogue) and through the operator-level backward (pixel-major kernel forced)."""
    import os

    import gsx  # noqa: F401
    from gsx import rasterizer, scenes
    sc = scenes.scene_small(seed=3, N=400)
    sc["width"], sc["height"] = 96, 64
    sc["K"] = scenes.intrinsics(70.0, 70.0, 48.0, 32.0)
    sc["scales"][:12] = torch.tensor([[0.3, 3e-8, 3e-8], [2e-8, 0.4, 2e-8], [1e-8, 1e-8, 0.5], [0.2, 1e-12, 1e-12]]).repeat(3, 1)
    sc["opacities"][:12] = 0.8
    sc["means"][:12, :2] *= 0.3
    cam = rasterizer.Camera(viewmat=scenes.look_at_viewmat((0.2, -0.1, -0.3), (0.0, 0.0, 2.5)).to(DEV), K=sc["K"].to(DEV), width=96, height=64)
    for bwd in ("gq", "pm"):
        os.environ["GSX_BWD"] = bwd
        try:
            model = scenes.to_splat_data(sc, DEV)
            for p in model.params():
                p.requires_grad_(True)
            out = rasterizer.rasterize_fused(cam, model, sc["background"].to(DEV))
            (out.render_hwc.sum() + out.alpha.sum()).backward()
            torch.cuda.synchronize()
        finally:
            os.environ.pop("GSX_BWD", None)
        assert bool(torch.isfinite(out.render_hwc).all())
        for n in ("means", "sh", "scaling_raw", "rotation_raw", "opacity_raw"):
            g = getattr(model, n).grad
            assert bool(torch.isfinite(g).all()), (bwd, n, int((~torch.isfinite(g)).sum()))
