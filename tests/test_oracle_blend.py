"""CPU cross-validation of the parts of the oracle the reference does not pin (UT projection, blend fwd/bwd):
float64 re-evaluation, an independently written differentiable torch forward + autograd, and sanity vs the
reference's EWA projection (torch_impl, different algorithm: only a bound)."""
import numpy as np
import pytest
import torch

from gsx import scenes  # noqa: F401  (package alias)
from tests.helpers import oracle_pipeline, rel_l2


def _small(N=400, size=64, seed=3):
    sc = scenes.scene_small(seed=seed, N=N)
    sc["width"] = sc["height"] = size
    sc["K"] = scenes.intrinsics(50.0, 50.0, size / 2.0, size / 2.0)
    sc["background"] = torch.tensor([0.1, 0.2, 0.3])
    return sc


def test_f32_vs_f64_forward():
    sc = _small()
    o32 = oracle_pipeline(sc, np.float32, frag_rel=1e-3)
    # identical binning + colours so that only the blend arithmetic differs
    o64 = oracle_pipeline(sc, np.float64, isect_override=(o32["tile_offsets"], o32["flatten_ids"]),
                          colors_override=o32["colors"])
    ok = o32["fragile"] == 0
    assert ok.mean() > 0.95
    err = np.abs(o32["renders"].astype(np.float64) - o64["renders"])[ok]
    assert err.max() < 1e-4, err.max()
    assert np.abs(o32["alphas"].astype(np.float64) - o64["alphas"])[ok].max() < 1e-4


def _torch_forward(sc, colors, offsets, flat, dtype=torch.float64):
    """Independent differentiable restatement of the blend (per-pixel loop, tiny scenes only)."""
    means = sc["means"].to(dtype).requires_grad_(True)
    quats = sc["quats"].to(dtype).requires_grad_(True)
    scales = sc["scales"].to(dtype).requires_grad_(True)
    opac = sc["opacities"].to(dtype).requires_grad_(True)
    cols = torch.tensor(colors[0], dtype=dtype, requires_grad=True)
    W, H = sc["width"], sc["height"]
    K = sc["K"].to(dtype)
    vm = sc["viewmat"].to(dtype)
    Rinv = vm[:3, :3].T
    org = -Rinv @ vm[:3, 3]
    qn = quats / quats.norm(dim=-1, keepdim=True)
    w, x, y, z = qn.unbind(-1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                     2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                     2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)
    M = (1.0 / scales)[:, :, None] * R.transpose(1, 2)
    bg = sc["background"].to(dtype)
    img = torch.zeros(H, W, 3, dtype=dtype)
    alpha_img = torch.zeros(H, W, dtype=dtype)
    tw = (W + 15) // 16
    off = offsets.reshape(-1)
    n_is = flat.shape[0]
    for i in range(H):
        for j in range(W):
            tid = (i // 16) * tw + (j // 16)
            s = int(off[tid]); e = int(off[tid + 1]) if tid + 1 < off.shape[0] else n_is
            d = torch.stack([(j + 0.5 - K[0, 2]) / K[0, 0], (i + 0.5 - K[1, 2]) / K[1, 1], torch.tensor(1.0, dtype=dtype)])
            d = Rinv @ (d / d.norm())
            T = torch.tensor(1.0, dtype=dtype)
            c = torch.zeros(3, dtype=dtype)
            for k in range(s, e):
                g = int(flat[k])
                gro = M[g] @ (org - means[g])
                grd = M[g] @ d
                grd = grd / grd.norm()
                gc = torch.linalg.cross(grd, gro)
                a = torch.clamp(opac[g] * torch.exp(-0.5 * (gc * gc).sum()), max=0.999)
                if a.item() < 1 / 255:
                    continue
                nT = T * (1 - a)
                if nT.item() <= 1e-4:
                    break
                c = c + cols[g] * a * T
                T = nT
            img[i, j] = c + T * bg
            alpha_img[i, j] = 1 - T
    return img, alpha_img, (means, quats, scales, opac, cols)


def test_backward_vs_torch_autograd_f64():
    sc = _small(N=60, size=32, seed=5)
    sc["scales"] = sc["scales"] * 2.0
    rng = np.random.default_rng(0)
    H, W = sc["height"], sc["width"]
    v_rc = rng.standard_normal((1, H, W, 3))
    v_ra = rng.standard_normal((1, H, W, 1))
    o = oracle_pipeline(sc, np.float64, v_render_colors=v_rc, v_render_alphas=v_ra)
    img, alpha, leaves = _torch_forward(sc, o["colors"], o["tile_offsets"], o["flatten_ids"])
    np.testing.assert_allclose(img.detach().numpy(), o["renders"][0], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(alpha.detach().numpy(), o["alphas"][0, ..., 0], rtol=1e-9, atol=1e-9)
    loss = (img * torch.tensor(v_rc[0])).sum() + (alpha * torch.tensor(v_ra[0, ..., 0])).sum()
    grads = torch.autograd.grad(loss, leaves)
    names = ["v_means", "v_quats", "v_scales", "v_opacities", "v_colors"]
    for n, g in zip(names, grads):
        ref = g.numpy()
        got = o[n].reshape(ref.shape)
        assert rel_l2(got, ref) < 1e-8, (n, rel_l2(got, ref))


def test_backward_f32_vs_f64():
    sc = _small(N=300, size=48, seed=7)
    rng = np.random.default_rng(1)
    H, W = sc["height"], sc["width"]
    v_rc = rng.standard_normal((1, H, W, 3)).astype(np.float32)
    v_ra = rng.standard_normal((1, H, W, 1)).astype(np.float32)
    o32 = oracle_pipeline(sc, np.float32, v_render_colors=v_rc, v_render_alphas=v_ra)
    o64 = oracle_pipeline(sc, np.float64, v_render_colors=v_rc, v_render_alphas=v_ra,
                          isect_override=(o32["tile_offsets"], o32["flatten_ids"]), colors_override=o32["colors"])
    # feed the f64 backward the f32 forward's discrete state so both walk the same Gaussians
    for n in ["v_means", "v_quats", "v_scales", "v_colors", "v_opacities"]:
        assert rel_l2(o32[n], o64[n]) < 5e-3, (n, rel_l2(o32[n], o64[n]))


def test_projection_f32_vs_f64_and_vs_ewa_bound():
    from oracle import ref
    sc = _small(N=2000, size=256, seed=11)
    sc["K"] = scenes.intrinsics(200.0, 200.0, 128.0, 128.0)
    o32 = oracle_pipeline(sc, np.float32)
    o64 = oracle_pipeline(sc, np.float64)
    vis = (o32["radii"] > 0).all(-1) & (o64["radii"] > 0).all(-1)
    assert vis.mean() > 0.5
    assert ((o32["radii"] > 0).all(-1) != (o64["radii"] > 0).all(-1)).mean() < 5e-3
    assert np.abs(o32["means2d"][vis] - o64["means2d"][vis]).max() < 5e-2       # UT cancellation noise
    assert np.abs(o32["radii"][vis].astype(np.int64) - o64["radii"][vis]).max() <= 1
    assert np.abs(o32["depths"][vis] - o64["depths"][vis]).max() < 1e-5
    if ref.available():  # the reference's EWA projection: a different algorithm, means2d agree to ~0.1 px
        r, m2d, dep, con = ref.ewa_projection(sc["means"].numpy(), sc["quats"].numpy(), sc["scales"].numpy(),
                                              sc["viewmat"].numpy()[None], sc["K"].numpy()[None], 256, 256)
        both = vis & (r > 0).all(-1)
        assert np.abs(m2d[both] - o32["means2d"][both]).max() < 0.5
        assert np.abs(dep[both] - o32["depths"][both]).max() < 1e-4
