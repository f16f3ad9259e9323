"""The "next" rows of SURVEY §8f against the REFERENCE's own kernels executed on the same MI355X (oracle/_ref/gsplat_ref_train.so and
gsplat_ref_hip.so: its .cu files compiled unmodified for gfx950, oracle/build_ref_train.sh / build_ref_hip.sh):
  f1 fused Adam       fastgs/optimizer adam_step_cu (the kernel src/training/optimizers/fused_adam.cpp:20-96 calls per group)
  f2 photometric loss src/training/kernels/ssim.cu behind include/kernels/fused_ssim.cuh, composed as trainer.cpp:103-127 does
  f3 MCMC operators   gsplat relocation / add_noise / quats_to_rotmats (RelocationCUDA.cu, QuatToRotmatCUDA.cu)
Tolerances are stated per check; where both sides run the same fp32 operation sequence they are a few ulp."""
import math

import numpy as np
import pytest
import torch

from oracle import ref_hip

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _train():
    m = ref_hip.load_train()
    if m is None:
        pytest.skip("oracle/_ref/gsplat_ref_train.so not built (bash oracle/build_ref_train.sh, needs /root/reference)")
    return m


def _ref():
    m = ref_hip.load()
    if m is None:
        pytest.skip("oracle/_ref/gsplat_ref_hip.so not built")
    return m


@pytest.mark.parametrize("n", [1, 1000, (1 << 20) + 7])
def test_fused_adam_vs_reference_kernel(n):
    """Ten steps of gsx_adam_step against adam_step_cu on the same inputs: parameters and both moments agree to 2 ulp of fp32 per step
    (one FMA-contraction choice apart); after ten steps 1e-6 relative + 1e-7 of the largest entry."""
    import gsx  # noqa: F401
    from gsx import ops
    ref = _train()
    g = torch.Generator(device="cpu").manual_seed(n)
    p0 = torch.randn(n, generator=g)
    P, Pr = p0.to(DEV).clone(), p0.to(DEV).clone()
    M, V, Mr, Vr = (torch.zeros(n, device=DEV) for _ in range(4))
    lr, b1, b2, eps = 1.6e-4, 0.9, 0.999, 1e-15
    for step in range(1, 11):
        G = (torch.randn(n, generator=g) * (10.0 ** torch.randint(-6, 1, (n,), generator=g).float())).to(DEV)
        if step == 4:
            G.zero_()                                                # a step with no gradient still decays the moments
        bc1, bc2 = 1.0 / (1.0 - b1 ** step), 1.0 / np.sqrt(1.0 - b2 ** step)
        ops.adam_step(P, M, V, G, lr, b1, b2, eps, bc1, bc2)
        ref.adam_step(Pr, Mr, Vr, G, lr, b1, b2, eps, bc1, bc2)
        torch.cuda.synchronize()
        if step == 1:
            for a, b in ((P, Pr), (M, Mr), (V, Vr)):
                assert ((a - b).abs() <= 2.4e-7 * b.abs() + 1e-30).all()
    for a, b, name in ((P, Pr, "param"), (M, Mr, "exp_avg"), (V, Vr, "exp_avg_sq")):   # (exp_avg sums signed terms: entries near zero)
        bad = (a - b).abs() > 1e-6 * b.abs() + 1e-7 * b.abs().max()
        assert not bad.any(), (name, ((a - b).abs() / (b.abs() + 1e-7 * b.abs().max())).max().item())


@pytest.mark.parametrize("C,H,W", [(1, 64, 96), (2, 37, 53), (1, 9, 20), (1, 270, 480), (1, 1080, 1920)])
def test_photometric_loss_vs_reference_kernels(C, H, W):
    """loss.photometric_loss (one fused forward + one fused backward kernel) against the reference's composition executed with ITS
    kernels: clamp -> l1_loss -> 1 - fused_ssim(rendered, gt, "valid", train) -> (1 - lambda) L1 + lambda D-SSIM (trainer.cpp:103-127),
    autograd through fs_internal::_FusedSSIM.  Loss to 2e-6 absolute, gradient to 1e-4 of its largest entry (the 1e-3 rel-L2 bar of
    north_star with margin)."""
    import gsx  # noqa: F401
    from gsx import loss
    ref = _train()
    rng = np.random.default_rng(H * W)
    r = (rng.random((C, H, W, 3), dtype=np.float32) * 1.4 - 0.2).astype(np.float32)
    gt = rng.random((C, 3, H, W), dtype=np.float32)
    R = torch.from_numpy(r).to(DEV).requires_grad_(True)
    G = torch.from_numpy(gt).to(DEV)
    val, parts = loss.photometric_loss(R, G, 0.2, return_parts=True)
    val.backward()
    R2 = torch.from_numpy(r).to(DEV).requires_grad_(True)
    rendered = R2.clamp(0, 1).permute(0, 3, 1, 2)
    l1 = torch.nn.functional.l1_loss(rendered, G)
    ssim = ref.fused_ssim(rendered, G, "valid", True)
    want = (1.0 - 0.2) * l1 + 0.2 * (1.0 - ssim)
    want.backward()
    assert abs(val.item() - want.item()) < 2e-6 and abs(parts[1].item() - l1.item()) < 2e-6 and abs(parts[2].item() - ssim.item()) < 2e-6
    gmax = R2.grad.abs().max().item()
    assert (R.grad - R2.grad).abs().max().item() <= 1e-4 * gmax
    rel_l2 = ((R.grad - R2.grad).norm() / R2.grad.norm()).item()
    assert rel_l2 < 1e-4, rel_l2
    # the standalone fused SSIM operators (ops.fusedssim / fusedssim_backward = the reference's kernel interface, kernels/ssim.cuh)
    from gsx import ops
    A, B = rendered.detach().contiguous(), G
    m, d0, d1, d2 = ops.fusedssim(1e-4, 9e-4, A, B, True)
    mr, r0, r1, r2 = ref.fusedssim(1e-4, 9e-4, A, B, True)
    for a, b in ((m, mr), (d0, r0), (d1, r1), (d2, r2)):
        assert (a - b).abs().max().item() <= 2e-5 * max(1.0, b.abs().max().item())
    up = torch.from_numpy(rng.standard_normal(tuple(A.shape)).astype(np.float32)).to(DEV)
    g = ops.fusedssim_backward(1e-4, 9e-4, A, B, up, d0, d1, d2)
    gr = ref.fusedssim_backward(1e-4, 9e-4, A, B, up, r0, r1, r2)
    assert (g - gr).abs().max().item() <= 1e-4 * gr.abs().max().item()


def test_mcmc_operators_vs_reference_kernels():
    """relocation (MCMC eq. 9, binomial table of n_max = 51), add_noise (opacity-gated covariance noise) and quats_to_rotmats against
    gsplat's own kernels."""
    import gsx  # noqa: F401
    from gsx import ops
    ref = _ref()
    rng = np.random.default_rng(11)
    N, n_max = 20011, 51
    opac = torch.from_numpy(rng.uniform(0.005, 0.995, N).astype(np.float32)).to(DEV)
    scales = torch.from_numpy(np.exp(rng.uniform(-6, 1, (N, 3))).astype(np.float32)).to(DEV)
    ratios = torch.from_numpy(rng.integers(1, n_max + 1, N).astype(np.int32)).to(DEV)
    binoms = torch.zeros(n_max, n_max)
    for n in range(n_max):
        for k in range(n + 1):
            binoms[n, k] = float(math.comb(n, k))
    binoms = binoms.to(DEV)
    o, s = ops.relocation(opac, scales, ratios, binoms, n_max)
    orf, srf = ref.relocation(opac, scales, ratios, binoms, n_max)
    assert ((o - orf).abs() <= 2e-6 * orf.abs() + 1e-9).all(), (o - orf).abs().max().item()
    assert ((s - srf).abs() <= 2e-5 * srf.abs() + 1e-12).all(), ((s - srf).abs() / srf.abs()).max().item()
    raw_o = torch.from_numpy(rng.standard_normal(N).astype(np.float32) * 3).to(DEV)
    raw_s = torch.from_numpy((rng.standard_normal((N, 3)) * 0.5 - 3).astype(np.float32)).to(DEV)
    raw_q = torch.from_numpy(rng.standard_normal((N, 4)).astype(np.float32)).to(DEV)
    noise = torch.from_numpy(rng.standard_normal((N, 3)).astype(np.float32)).to(DEV)
    mu = torch.from_numpy(rng.standard_normal((N, 3)).astype(np.float32)).to(DEV)
    a, b = mu.clone(), mu.clone()
    ops.add_noise(raw_o, raw_s, raw_q, noise, a, 0.05)
    ref.add_noise(raw_o, raw_s, raw_q, noise, b, 0.05)
    torch.cuda.synchronize()
    step = (b - mu).abs().max().item()
    assert step > 0 and (a - b).abs().max().item() <= 2e-6 * max(step, 1.0), ((a - b).abs().max().item(), step)
    Rm, Rr = ops.quats_to_rotmats(raw_q), ref.quats_to_rotmats(raw_q)
    assert (Rm - Rr).abs().max().item() <= 1e-6   # (measured 6e-7: the normalisation is a reciprocal square root on one side)


def _training_iterations_vs_reference_chain(sc, K, vms, targets, bg, its, label, update_tol, loss_tol):
    """The body of the two tests below: `its` training iterations of the bench path on scene `sc` (cameras vms, in rotation) against the same iterations
    composed from the reference's own kernels."""
    import gsx  # noqa: F401
    from gsx import loss as gloss
    from gsx import optim, rasterizer, scenes
    ref, rt = _ref(), _train()
    N, W, H, deg = sc["means"].shape[0], sc["width"], sc["height"], sc["sh_degree"]
    n_cam = len(vms)
    lrs = {"means": 1.6e-4, "sh0": 2.5e-3, "shN": 2.5e-3 / 20.0, "scaling": 5e-3, "rotation": 1e-3, "opacity": 5e-2}
    b1, b2, eps = 0.9, 0.999, 1e-15

    # ---- ours
    model = scenes.to_splat_data(sc, DEV)
    start = [p.detach().clone() for p in model.params()]
    for p in model.params():
        p.requires_grad_(True)
    from gsx import distributed as gdist
    bucket = gdist.GradBucket(model.params())
    sinks = bucket.sinks()
    opt = optim.FusedAdam.for_splat_data(model)
    for name in ("means", "sh0", "shN", "scaling", "rotation", "opacity"):   # as if the run had reached the iteration before the first one
        opt.state["step:" + name] = its[0] - 1
    losses = []
    for k, it in enumerate(its):
        cam = rasterizer.Camera(viewmat=vms[k % n_cam], K=K, width=W, height=H)
        sinks["_sh_adam"] = opt.begin_fused_sh_step(it)
        out = rasterizer.rasterize_fused(cam, model, bg, grad_sinks=sinks, guarded=True)
        l = gloss.photometric_loss(out.render_hwc, targets[k % n_cam], 0.2)
        gloss.backward(l)
        opt.step(it, skip_sh=sinks["_sh_adam"] is not None)
        losses.append(float(l))
    ours = [p.detach().clone() for p in model.params()]   # means, sh, scaling_raw, rotation_raw, opacity_raw

    # ---- the reference's kernels, composed as its trainer does
    P = {"means": start[0].clone(), "sh0": start[1][:, :1].contiguous().clone(), "shN": start[1][:, 1:].contiguous().clone(),
         "scaling": start[2].clone(), "rotation": start[3].clone(), "opacity": start[4].clone()}
    M = {n: torch.zeros_like(p) for n, p in P.items()}
    V = {n: torch.zeros_like(p) for n, p in P.items()}
    steps = {n: its[0] - 1 for n in P}
    ref_losses = []
    for k, it in enumerate(its):
        vm = vms[k % n_cam][None].contiguous()
        raw = {n: p.detach().clone().requires_grad_(True) for n, p in P.items()}
        scales, quats = torch.exp(raw["scaling"]), torch.nn.functional.normalize(raw["rotation"], dim=-1)
        opac = torch.sigmoid(raw["opacity"]).squeeze(-1)
        shs = torch.cat([raw["sh0"], raw["shN"]], 1)
        R = ref_hip.render_chain(ref, raw["means"].detach(), quats.detach().contiguous(), scales.detach().contiguous(), opac.detach().contiguous(),
                                 shs.detach().contiguous(), deg, vm, K[None].contiguous(), W, H, bg[None].contiguous())
        ren = R["renders"].detach().requires_grad_(True)
        rendered = ren.clamp(0, 1).permute(0, 3, 1, 2)
        gt = targets[k % n_cam][None]
        lo = (1.0 - 0.2) * torch.nn.functional.l1_loss(rendered, gt) + 0.2 * (1.0 - rt.fused_ssim(rendered, gt, "valid", True))
        lo.backward()
        ref_losses.append(float(lo))
        op1 = opac.detach()[None].contiguous()
        gb = ref.rasterize_to_pixels_from_world_3dgs_bwd(raw["means"].detach(), quats.detach().contiguous(), scales.detach().contiguous(), R["colors"], op1,
                                                         bg[None].contiguous(), None, W, H, 16, vm, None, K[None].contiguous(), 0, None, 4, None, None, None,
                                                         R["tile_offsets"], R["flatten_ids"], R["alphas"], R["last_ids"], ren.grad.contiguous(),
                                                         torch.zeros_like(R["alphas"]))
        v_means, v_quats, v_scales, v_colors, v_opac = gb
        v_col = (v_colors * (R["colors"] > 0)).contiguous()                      # clamp_min(c + 0.5, 0)
        v_coeffs, v_dirs = ref.spherical_harmonics_bwd(16, deg, R["dirs"].contiguous(), shs.detach()[None].contiguous(), R["masks"], v_col, True)
        torch.autograd.backward([scales, quats, opac], [v_scales, v_quats, v_opac[0]])
        grads = {"means": v_means + v_dirs[0], "sh0": v_coeffs[0][:, :1].contiguous(), "shN": v_coeffs[0][:, 1:].contiguous(),
                 "scaling": raw["scaling"].grad, "rotation": raw["rotation"].grad, "opacity": raw["opacity"].grad}
        for i, n in enumerate(("means", "sh0", "shN", "scaling", "rotation", "opacity"), start=1):   # fused_adam.cpp:20-96
            steps[n] += 1
            if i == 3 and it <= 1000:
                continue                                                         # shN: counter advances, no update (fused_adam.cpp:66-70)
            t = steps[n]
            rt.adam_step(P[n].view(-1), M[n].view(-1), V[n].view(-1), grads[n].contiguous().view(-1), lrs[n], b1, b2, eps,
                         1.0 / (1.0 - b1 ** t), 1.0 / math.sqrt(1.0 - b2 ** t))
    theirs = [P["means"], torch.cat([P["sh0"], P["shN"]], 1), P["scaling"], P["rotation"], P["opacity"]]
    torch.cuda.synchronize()
    assert max(abs(a - b) for a, b in zip(losses, ref_losses)) < loss_tol, (losses, ref_losses)
    rec = {}
    for n, s0, a, b in zip(("means", "sh", "scaling_raw", "rotation_raw", "opacity_raw"), start, ours, theirs):
        da, db = (a - s0).float(), (b.reshape(s0.shape) - s0).float()
        rec[n] = float((da - db).norm() / db.norm())
        assert float(db.norm()) > 0.0
    from tests.helpers import parity_record
    parity_record("%s: gsx bench path vs the reference's kernels composed as its trainer does — rel-L2 of the accumulated parameter UPDATES; max |loss "
                  "difference| over the iterations" % label, max_loss_diff=max(abs(a - b) for a, b in zip(losses, ref_losses)), **rec)
    for n, v in rec.items():
        assert v < update_tol, (n, v, rec)   # (Adam normalises: an element whose gradient is rounding noise moves by +-lr whatever its size)


def test_training_iterations_vs_the_reference_chain():
    """END TO END: ten training iterations of the path bench.py times — rasterize_fused (fused front end, guarded binned intersection, packed-record
    blend, Gaussian-major backward, fused SH backward + the SH tensor's Adam step) + the fused photometric loss + FusedAdam — against the same
    iterations composed from the REFERENCE's own kernels the way its trainer composes them (trainer.cpp:579-800): projection_ut / SH / intersect /
    blend forward and backward (gsplat/*.cu), fused SSIM (ssim.cu), torch glue for the activations / clamp / L1 as upstream, adam_step_cu per parameter
    group with the lrs and quirks of strategy_utils.cpp:35-40 / fused_adam.cpp:20-96 (iterations 996..1005: shN frozen up to iteration 1000 while its
    step counter advances).  The per-iteration losses and the accumulated parameter updates must agree."""
    import gsx  # noqa: F401
    from gsx import scenes
    N, W, H, deg = 3000, 128, 96, 3
    sc = scenes.scene_small(seed=31, N=N)
    g = torch.Generator().manual_seed(4)
    sc["sh"] = (torch.rand(N, 16, 3, generator=g) - 0.5) * 0.6
    sc["sh_degree"] = deg
    sc["width"], sc["height"] = W, H
    K = scenes.intrinsics(100.0, 100.0, W / 2.0, H / 2.0).to(DEV)
    vms = [scenes.look_at_viewmat(e, (0.0, 0.0, 2.5)).to(DEV) for e in ((0.3, 0.0, -0.6), (-0.3, 0.2, -0.5), (0.0, -0.3, -0.7))]
    targets = [torch.rand(3, H, W, generator=g).to(DEV) for _ in vms]
    bg = (sc["background"] + 0.1).to(DEV)
    _training_iterations_vs_reference_chain(sc, K, vms, targets, bg, list(range(996, 1006)), "ten training iterations (996..1005, 3000 Gaussians @128x96, 3 cameras)", 2e-2, 2e-5)


def test_s1m_training_iterations_vs_the_reference_chain():
    """The same at the BASELINE scale: six training iterations (998 .. 1003: across the end of the shN freeze) of S-1M @ 1920 x 1080 through three of the bench's
    camera poses against the reference's kernels composed as its trainer composes them (one reference iteration costs ~30 ms on this GPU)."""
    import gsx  # noqa: F401
    from gsx import scenes
    sc = scenes.scene_1m()
    W, H = sc["width"], sc["height"]
    K = sc["K"].to(DEV)
    zc = float(sc["means"][:, 2].mean())
    vms = [sc["viewmat"].to(DEV)] + [scenes.look_at_viewmat(e, (0.0, 0.0, zc)).to(DEV) for e in ((0.4, 0.0, 0.0), (-0.2828, 0.2828, -0.5))]
    g = torch.Generator().manual_seed(4)
    targets = [torch.rand(3, H, W, generator=g).to(DEV) for _ in vms]
    bg = sc["background"].to(DEV)
    _training_iterations_vs_reference_chain(sc, K, vms, targets, bg, list(range(998, 1004)), "six training iterations (998..1003) of S-1M @1080p, 3 cameras", 3e-2, 2e-5)
