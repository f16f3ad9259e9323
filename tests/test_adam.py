"""Fused Adam: the oracle restates the reference kernel (adam_kernels.cuh:13-38) and is checked against
torch.optim.Adam on CPU; the HIP kernel is checked against the oracle on the GPU, dense and row-strided (sh0 / shN views);
FusedAdam keeps the reference's shN quirks."""
import numpy as np
import pytest
import torch

from oracle import oracle


def test_oracle_adam_matches_torch_adam():
    torch.manual_seed(0)
    p = torch.randn(257, 3, dtype=torch.float64)
    ref = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-2, betas=(0.9, 0.999), eps=1e-8)
    pn, m, v = p.numpy().copy(), np.zeros_like(p.numpy()), np.zeros_like(p.numpy())
    for step in range(1, 6):
        g = torch.randn_like(p)
        ref.grad = g.clone()
        opt.step()
        pn, m, v = oracle.adam_step(pn, m, v, g.numpy(), 1e-2, 0.9, 0.999, 1e-8, step)
        np.testing.assert_allclose(pn, ref.detach().numpy(), rtol=1e-10, atol=1e-12)


@pytest.mark.gpu
def test_gpu_adam_dense_and_strided_vs_oracle():
    import gsx  # noqa: F401
    from gsx import ops
    dev = "cuda:0"
    rng = np.random.default_rng(0)
    for shape in [(1000, 3), (4099,), (513, 16, 3)]:
        p, g = rng.standard_normal(shape).astype(np.float32), rng.standard_normal(shape).astype(np.float32)
        m, v = (rng.random(shape) * 0.1).astype(np.float32), (rng.random(shape) * 0.01).astype(np.float32)
        P, M, V, G = (torch.from_numpy(a.copy()).to(dev) for a in (p, m, v, g))
        bc1, bc2 = 1.0 / (1 - 0.9 ** 3), 1.0 / np.sqrt(1 - 0.999 ** 3)
        if len(shape) == 1:   # the reference's own entry point (fast_gs::optimizer::adam_step_wrapper, adam_api.h:11-21)
            ops.adam_step_wrapper(P, M, V, G, 1e-2, 0.9, 0.999, 1e-8, bc1, bc2)
        else:
            ops.adam_step(P, M, V, G, 1e-2, 0.9, 0.999, 1e-8, bc1, bc2)
        rp, rm, rv = oracle.adam_step(p, m, v, g, 1e-2, 0.9, 0.999, 1e-8, 3)
        np.testing.assert_allclose(P.cpu().numpy(), rp, rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(M.cpu().numpy(), rm, rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(V.cpu().numpy(), rv, rtol=1e-6, atol=1e-9)
    # row-strided views: the shN block of an [N,K,3] tensor
    sh, gsh = rng.standard_normal((300, 16, 3)).astype(np.float32), rng.standard_normal((300, 16, 3)).astype(np.float32)
    SH, GSH = torch.from_numpy(sh.copy()).to(dev), torch.from_numpy(gsh).to(dev)
    M, V = torch.zeros(300, 15, 3, device=dev), torch.zeros(300, 15, 3, device=dev)
    ops.adam_step(SH[:, 1:], M, V, GSH[:, 1:], 1e-2, 0.9, 0.999, 1e-8, 10.0, 1.0 / np.sqrt(1 - 0.999))
    rp, rm, rv = oracle.adam_step(sh[:, 1:], np.zeros((300, 15, 3), np.float32), np.zeros((300, 15, 3), np.float32), gsh[:, 1:], 1e-2,
                                  0.9, 0.999, 1e-8, 1)
    out = SH.cpu().numpy()
    np.testing.assert_allclose(out[:, 1:], rp, rtol=1e-5, atol=1e-7)
    assert np.array_equal(out[:, :1], sh[:, :1])           # the sh0 block is untouched


@pytest.mark.gpu
def test_fused_adam_reference_quirks():
    import gsx  # noqa: F401
    from gsx import optim, scenes
    dev = "cuda:0"
    sc = scenes.scene_small(seed=1, N=200)
    g = torch.Generator().manual_seed(0)
    sc["sh"] = torch.rand(200, 16, 3, generator=g)
    sc["sh_degree"] = 3
    model = scenes.to_splat_data(sc, dev)
    for p in model.params():
        p.requires_grad_(True)
        p.grad = torch.ones_like(p)
    opt = optim.FusedAdam.for_splat_data(model)
    before = [p.detach().clone() for p in model.params()]
    opt.step(iteration=1)
    # every group moved by ~lr (first Adam step = lr * sign(g)) except shN, frozen for the first 1000 iterations
    assert torch.allclose(model.means, before[0] - 0.00016, atol=1e-7)
    assert torch.allclose(model.sh[:, :1], before[1][:, :1] - 0.0025, atol=1e-6)
    assert torch.equal(model.sh[:, 1:], before[1][:, 1:])
    assert opt.step_count("shN") == 1                       # ... although its step counter advanced (fused_adam.cpp:66-70)
    opt.step(iteration=1001)
    assert not torch.equal(model.sh[:, 1:], before[1][:, 1:])
    sched = optim.ExponentialLR(opt, 0.5, 0)
    sched.step()
    # (the group learning rates are the reference's C floats widened to double: 0.00016f = 0.00015999999595806003, strategy_utils.cpp:35-40)
    assert abs(opt.groups[0]["lr"] - 0.5 * float(np.float32(0.00016))) < 1e-15 and opt.groups[1]["lr"] == float(np.float32(0.0025))


@pytest.mark.gpu
def test_gpu_adam_split_blocks_vs_oracle():
    import gsx  # noqa: F401
    from gsx import ops
    rng = np.random.default_rng(3)
    sh, g = rng.standard_normal((777, 16, 3)).astype(np.float32), rng.standard_normal((777, 16, 3)).astype(np.float32)
    m, v = (rng.random((777, 16, 3)) * 0.1).astype(np.float32), (rng.random((777, 16, 3)) * 0.01).astype(np.float32)
    bc1, bc2 = 1.0 / (1 - 0.9 ** 5), 1.0 / np.sqrt(1 - 0.999 ** 5)
    for do_b in (True, False):
        P, M, V, G = (torch.from_numpy(a.copy()).cuda() for a in (sh, m, v, g))
        ops.adam_step_split(P, M, V, G, 3, 2.5e-3, 1.25e-4, True, do_b, 0.9, 0.999, 1e-8, bc1, bc2)
        p0, m0, v0 = oracle.adam_step(sh[:, :1], m[:, :1], v[:, :1], g[:, :1], 2.5e-3, 0.9, 0.999, 1e-8, 5)
        np.testing.assert_allclose(P.cpu().numpy()[:, :1], p0, rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(M.cpu().numpy()[:, :1], m0, rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(V.cpu().numpy()[:, :1], v0, rtol=1e-6, atol=1e-9)
        if do_b:
            p1, m1, v1 = oracle.adam_step(sh[:, 1:], m[:, 1:], v[:, 1:], g[:, 1:], 1.25e-4, 0.9, 0.999, 1e-8, 5)
            np.testing.assert_allclose(P.cpu().numpy()[:, 1:], p1, rtol=1e-5, atol=1e-7)
            np.testing.assert_allclose(V.cpu().numpy()[:, 1:], v1, rtol=1e-6, atol=1e-9)
        else:
            assert np.array_equal(P.cpu().numpy()[:, 1:], sh[:, 1:]) and np.array_equal(M.cpu().numpy()[:, 1:], m[:, 1:])


@pytest.mark.gpu
@pytest.mark.parametrize("K", [16, 9])
def test_fused_adam_matches_torch_adam_over_iterations(K):
    """Whole-optimizer check incl. the split SH launch (K = 16) and the strided fallback (K = 9) against torch.optim.Adam."""
    import gsx  # noqa: F401
    from gsx import optim, scenes
    sc = scenes.scene_small(seed=2, N=150)
    gen = torch.Generator().manual_seed(1)
    sc["sh"] = torch.rand(150, K, 3, generator=gen)
    sc["sh_degree"] = {16: 3, 9: 2}[K]
    model = scenes.to_splat_data(sc, "cuda:0")
    params = model.params()
    for p in params:
        p.requires_grad_(True)
    ref = [p.detach().cpu().double().requires_grad_(True) for p in params]
    ref_sh0, ref_shN = ref[1][:, :1].detach().clone().requires_grad_(True), ref[1][:, 1:].detach().clone().requires_grad_(True)
    lrs = [0.00016, 0.0025, 0.0025 / 20, 0.005, 0.001, 0.05]
    topt = torch.optim.Adam([{"params": [t], "lr": lr} for t, lr in zip([ref[0], ref_sh0, ref_shN, ref[2], ref[3], ref[4]], lrs)],
                            betas=(0.9, 0.999), eps=1e-15)
    opt = optim.FusedAdam.for_splat_data(model)
    for it in range(1001, 1006):                      # past the shN freeze
        for p in params:
            p.grad = torch.randn(p.shape, generator=gen).to(p.device)
        for t, src in zip([ref[0], ref_sh0, ref_shN, ref[2], ref[3], ref[4]],
                          [params[0].grad, params[1].grad[:, :1], params[1].grad[:, 1:], params[2].grad, params[3].grad, params[4].grad]):
            t.grad = src.double().cpu()
        opt.step(it)
        topt.step()
    got_sh = params[1].detach().cpu().double()
    assert (params[0].detach().cpu().double() - ref[0].detach()).abs().max() < 1e-6
    assert (got_sh[:, :1] - ref_sh0.detach()).abs().max() < 1e-5 and (got_sh[:, 1:] - ref_shN.detach()).abs().max() < 1e-6
    for a, b in zip(params[2:], ref[2:]):
        assert (a.detach().cpu().double() - b.detach()).abs().max() < 1e-4


def test_begin_fused_sh_step_bookkeeping():
    """optim.FusedAdam.begin_fused_sh_step hands the sh0 / shN groups' step to the render backward: step counters advance exactly as in
    step(), the shN block is disabled during the first 1000 iterations (fused_adam.cpp:66-70) while its counter advances, the step sizes
    carry the bias correction of the advanced count; K * 3 not a multiple of 4 falls back (None).  No kernel runs: CPU tensors."""
    import math

    import gsx  # noqa: F401
    from gsx import optim, rasterizer

    def model(K):
        n = 5
        return rasterizer.SplatData(means=torch.zeros(n, 3), sh=torch.zeros(n, K, 3), scaling_raw=torch.zeros(n, 3), rotation_raw=torch.zeros(n, 4),
                                    opacity_raw=torch.zeros(n, 1), active_sh_degree=0)
    opt = optim.FusedAdam.for_splat_data(model(16))
    a = opt.begin_fused_sh_step(10)
    exp_avg, exp_avg_sq, step0, stepN, do0, doN, b1, b2, eps, bc2 = a
    assert exp_avg.shape == (5, 16, 3) and exp_avg_sq.shape == (5, 16, 3)
    assert do0 is True and doN is False                       # shN warm-up: untouched ...
    assert opt.step_count("sh0") == 0 and opt.step_count("shN") == 0   # counters are committed only once the fused kernel has run:
    opt.step(10, skip_sh=True)                                          # ... by the step() that follows the render backward
    assert opt.step_count("sh0") == 1 and opt.step_count("shN") == 1   # (the shN counter advances during its warm-up, fused_adam.cpp:66-70)
    lr0, lrN = float(np.float32(0.0025)), float(np.float32(0.0025) / np.float32(20.0))   # C floats widened to double, as upstream forms them
    assert abs(step0 - lr0 / (1 - 0.9)) < 1e-12 and abs(stepN - lrN / (1 - 0.9)) < 1e-12
    assert abs(bc2 - 1 / math.sqrt(1 - 0.999)) < 1e-9 and (b1, b2, eps) == (0.9, 0.999, 1e-15)
    a = opt.begin_fused_sh_step(1500)
    opt.step(1500, skip_sh=True)
    assert a[4] is True and a[5] is True and opt.step_count("shN") == 2
    assert abs(a[2] - lr0 / (1 - 0.9 ** 2)) < 1e-12
    assert opt.step_count("means") == 0                       # the other groups are stepped by step(skip_sh=True)
    assert optim.FusedAdam.for_splat_data(model(9)).begin_fused_sh_step(1500) is None   # 27 floats per row: no 16 B vectors
    import pytest
    with pytest.raises(RuntimeError):
        optim.FusedAdam.for_splat_data(model(16)).step(5, skip_sh=True)                 # nothing was handed to a render backward
    m = model(16)
    m.sh = torch.zeros(16, 5, 3).permute(1, 0, 2)                                       # non-contiguous SH tensor: no fused step
    assert optim.FusedAdam.for_splat_data(m).begin_fused_sh_step(1500) is None
