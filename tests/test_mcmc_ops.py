"""relocation / add_noise / quats_to_rotmats (gsplat/Ops.h:45-65): the oracle against closed forms on CPU, and the HIP
kernels against the oracle on the GPU.  Inputs follow the reference's tests/test_gsplat_ops.cpp:19-63."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _binoms(n_max=51):
    b = np.zeros((n_max, n_max), np.float32)
    for n in range(n_max):
        for k in range(n + 1):
            b[n, k] = math.comb(n, k)
    return b


def _inputs(N=100, seed=42):
    rng = np.random.default_rng(seed)
    opac = (rng.random(N) * 0.8 + 0.1).astype(np.float32)
    scales = (rng.random((N, 3)) * 0.5 + 0.1).astype(np.float32)
    ratios = rng.integers(1, 10, N).astype(np.int32)
    return opac, scales, ratios


def test_oracle_relocation_closed_form():
    opac, scales, ratios = _inputs()
    b = _binoms()
    new_o, new_s = oracle.relocation(opac.astype(np.float64), scales.astype(np.float64), ratios, b.astype(np.float64), 51)
    # the reference test's sanity checks (test_gsplat_ops.cpp:54-62) ...
    assert not np.isnan(new_o).any() and not np.isnan(new_s).any()
    assert (new_o >= 0).all() and (new_o <= 1).all() and (new_s > 0).all()
    # ... and Eq. (9): (1 - o_new)^n == 1 - o_old ;  ratio == 1 leaves the Gaussian unchanged
    np.testing.assert_allclose((1 - new_o) ** ratios, 1 - opac, rtol=1e-6)
    one = ratios == 1
    np.testing.assert_allclose(new_s[one], scales[one], rtol=1e-6)
    new_o32, new_s32 = oracle.relocation(opac, scales, ratios, b, 51)
    np.testing.assert_allclose(new_o32, new_o, rtol=1e-5)
    np.testing.assert_allclose(new_s32, new_s, rtol=1e-4)


def test_oracle_add_noise_vs_numpy():
    rng = np.random.default_rng(1)
    N = 64
    ro, rs = rng.standard_normal(N), rng.standard_normal((N, 3)) * 0.3 - 2
    rq, nz, mu = rng.standard_normal((N, 4)), rng.standard_normal((N, 3)), rng.standard_normal((N, 3))
    out = oracle.add_noise(ro, rs, rq, nz, mu, 0.01)
    q = rq / np.linalg.norm(rq, axis=1, keepdims=True)
    w, x, y, z = q.T
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z), 1 - 2 * (x * x + z * z),
                  2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).reshape(N, 3, 3)
    cov = R @ (np.exp(2 * rs)[:, :, None] * R.transpose(0, 2, 1))
    o = 1 / (1 + np.exp(-ro))
    f = 0.01 / (1 + np.exp(100 * o - 0.5))
    np.testing.assert_allclose(out, mu + f[:, None] * np.einsum("nij,nj->ni", cov, nz), rtol=1e-10, atol=1e-12)


@pytest.mark.gpu
def test_gpu_mcmc_ops_vs_oracle():
    import gsx  # noqa: F401
    from gsx import ops
    dev = "cuda:0"
    opac, scales, ratios = _inputs(N=1000)
    b = _binoms()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    new_o, new_s = ops.relocation(t(opac), t(scales), t(ratios), t(b), 51)
    ro, rs_ = oracle.relocation(opac, scales, ratios, b, 51)
    np.testing.assert_allclose(new_o.cpu().numpy(), ro, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(new_s.cpu().numpy(), rs_, rtol=2e-4, atol=1e-6)
    rng = np.random.default_rng(2)
    N = 5000
    ro_, rsc = rng.standard_normal(N).astype(np.float32), (rng.standard_normal((N, 3)) * 0.3 - 2).astype(np.float32)
    rq, nz, mu = (rng.standard_normal(s).astype(np.float32) for s in ((N, 4), (N, 3), (N, 3)))
    means = t(mu)
    ops.add_noise(t(ro_), t(rsc), t(rq), t(nz), means, 0.01)
    np.testing.assert_allclose(means.cpu().numpy(), oracle.add_noise(ro_, rsc, rq, nz, mu, 0.01), rtol=1e-4, atol=1e-6)
    g = np.load(os.path.join(GOLDEN, "quat_torch_impl.npz"))
    R = ops.quats_to_rotmats(t(g["quats"]))
    np.testing.assert_allclose(R.cpu().numpy(), g["rotmats"], rtol=1e-5, atol=1e-5)   # the reference's own quat_to_rotmat
    assert ops.relocation(t(opac[:0]), t(scales[:0]), t(ratios[:0]), t(b), 51)[0].numel() == 0


def test_morton_order_and_spatial_reorder_keep_the_model_as_a_set():
    """gsx.layout.morton_order is a permutation that clusters neighbours, and MCMC.reorder_spatially applies it to the parameters AND the
    Adam moments (CPU tensors: no kernel runs)."""
    import numpy as np
    import torch
    import gsx  # noqa: F401
    from gsx import layout, parameters, rasterizer, strategy
    g = torch.Generator().manual_seed(0)
    n = 4096
    means = torch.rand(n, 3, generator=g)
    order = layout.morton_order(means)
    assert torch.equal(torch.sort(order).values, torch.arange(n))
    # neighbours in memory are neighbours in space: mean distance between consecutive Gaussians drops by a large factor
    d_rand = (means[1:] - means[:-1]).norm(dim=1).mean()
    ms = means[order]
    d_sorted = (ms[1:] - ms[:-1]).norm(dim=1).mean()
    assert float(d_sorted) < 0.25 * float(d_rand)
    assert torch.equal(layout.morton_order(means), order)               # deterministic (stable sort): identical on every rank
    bad = means.clone()
    bad[7] = float("nan")
    assert torch.equal(torch.sort(layout.morton_order(bad)).values, torch.arange(n))   # non-finite rows do not break it
    model = rasterizer.SplatData(means=means.clone(), sh=torch.rand(n, 4, 3, generator=g), scaling_raw=torch.rand(n, 3, generator=g),
                                 rotation_raw=torch.rand(n, 4, generator=g), opacity_raw=torch.rand(n, 1, generator=g), active_sh_degree=1)
    st = strategy.MCMC(model, parameters.OptimizationParameters())
    mom = st.optimizer._moments("means")
    mom["exp_avg"].copy_(means * 3.0)           # moments that identify their row
    before = {k: getattr(model, k).detach().clone() for k in ("means", "sh", "scaling_raw", "rotation_raw", "opacity_raw")}
    calls = []
    st.before_reindex = lambda: calls.append("merge")
    st.on_resize = lambda m: calls.append("resize")
    perm = st.reorder_spatially()
    assert calls == ["merge", "resize"] and torch.equal(perm, order)
    for k, v in before.items():
        assert torch.equal(getattr(model, k).detach(), v[perm]) and getattr(model, k).requires_grad
    assert torch.equal(st.optimizer._moments("means")["exp_avg"], model.means.detach() * 3.0)   # the moments moved with their rows
    assert np.isclose(float(model.means.detach().sum()), float(before["means"].sum()), rtol=1e-6)
