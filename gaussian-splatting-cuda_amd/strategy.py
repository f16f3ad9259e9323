"""MCMC densification strategy: the host logic of src/training/strategies/mcmc.cpp on the HIP ops
`relocation` and `add_noise` (csrc/gsx_mcmc.hip) and the fused Adam (optim.py).

Mirrors, function by function: multinomial sampling by opacity (mcmc.cpp:43-83), relocate_gs (:114-189: dead = opacity <=
min_opacity or degenerate quaternion; teleport onto opacity-sampled live Gaussians, Eq. (9) opacities / scales, moments of
the sampled Gaussians reset), add_new_gs (:191-340: grow by 5 % up to max_cap, clones appended, zero moments appended),
inject_noise (:342-366: lr_means * 5e5 scaled noise through the covariance), post_backward (:368-393), step (:395-402),
is_refining (:501-505), optimizer / scheduler set-up (:446-499).  SplatData keeps SH as one [N,K,3] tensor, so the sh0 / shN
copies are one indexed copy.

Multi-GPU (one camera per rank): every rank must take identical decisions — pass a `generator` seeded identically on all
ranks (the sampling and the noise are the only random draws); torch's multinomial / randn are deterministic given the
generator state and identical inputs."""
import torch

from . import ops, optim


from .parameters import OptimizationParameters  # noqa: E402,F401  (include/core/parameters.hpp:15-93 + the JSON parameter files)


class MCMC:
    NOISE_LR = 5e5   # mcmc.hpp:79
    N_MAX = 51       # binomial table (mcmc.cpp:459-473)

    def __init__(self, model, params: OptimizationParameters, scene_scale: float = 1.0, generator=None):
        self.model, self.params, self.generator = model, params, generator
        dev = model.means.device
        import numpy as np
        binoms = np.zeros((self.N_MAX, self.N_MAX), np.float32)   # the reference's fp32 product loop, term by term (mcmc.cpp:461-471)
        for n in range(self.N_MAX):
            for k in range(n + 1):
                b = np.float32(1.0)
                for i in range(k):
                    b = np.float32(b * (np.float32(n - i) / np.float32(i + 1)))
                binoms[n, k] = b
        self.binoms = torch.from_numpy(binoms).to(dev).contiguous()
        self.optimizer = optim.FusedAdam.for_splat_data(model, params.means_lr, params.shs_lr, params.scaling_lr, params.rotation_lr,
                                                        params.opacity_lr, scene_scale)
        self.scheduler = optim.ExponentialLR(self.optimizer, 0.01 ** (1.0 / params.iterations), 0)
        self.on_resize = None   # callback(model) after add_new_gs replaced the parameter tensors (e.g. rebuild the gradient bucket)
        self.before_reindex = None    # callback() before rows are permuted / removed (distributed.ShardedAdam.merge_moments: every rank needs complete moments first)
        self.spatial_reorder = True   # after a growth step (the tensors are re-created anyway) store the Gaussians in Morton order: layout.py
        self._grew = False      # add_new_gs replaced the parameters this iteration: they carry no gradient (see step)

    # ---- helpers ---------------------------------------------------------------------------------------------------------
    def is_refining(self, it):
        p = self.params
        return it < p.stop_refine and it > p.start_refine and it % p.refine_every == 0

    def _multinomial(self, weights, n):
        if weights.shape[0] <= (1 << 24):
            return torch.multinomial(weights, n, True, generator=self.generator)
        cdf = torch.cumsum(weights.double() / weights.double().sum(), 0)   # mcmc.cpp:50-82 (inverse-CDF sampling)
        u = torch.rand(n, dtype=torch.float64, device=weights.device, generator=self.generator)
        return torch.searchsorted(cdf, u).clamp_max(weights.shape[0] - 1)

    def _relocated(self, opacities, sampled, ratios):
        new_op, new_sc = ops.relocation(opacities.index_select(0, sampled).contiguous(),
                                        self.model.get_scaling().index_select(0, sampled).contiguous(), ratios.contiguous(), self.binoms,
                                        self.N_MAX)
        return new_op.clamp_(self.params.min_opacity, 1.0 - 1e-7), new_sc

    # ---- mcmc.cpp:114-189 ------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def relocate_gs(self):
        m = self.model
        opac = m.get_opacity()
        dead = (opac <= self.params.min_opacity) | ((m.rotation_raw * m.rotation_raw).sum(-1) < 1e-8)
        # A Gaussian with a non-finite parameter is dead too (upstream has no such case: its multinomial asserts on the NaN opacity and the run ends).  Its Adam
        # moments are reset with it.  Reported once per event on stderr: it should not happen, and when it does the pattern says where from.
        # (row sums: one pass over each tensor and an [N] result — NaN and +-inf survive a sum, and inf - inf is NaN; no [N, 48] boolean temporaries at 5 M Gaussians)
        finite = torch.isfinite(opac + m.means.data.sum(-1) + m.scaling_raw.data.sum(-1) + m.rotation_raw.data.sum(-1) + m.sh.data.reshape(m.sh.shape[0], -1).sum(-1))
        n_bad = int((~finite).sum())
        if n_bad:
            import sys
            bad_idx = (~finite).nonzero().squeeze(-1)
            per = {n: int((~torch.isfinite(getattr(m, n).data.reshape(m.means.shape[0], -1))).any(-1).sum()) for n in ("means", "sh", "scaling_raw", "rotation_raw", "opacity_raw")}
            runs = int((bad_idx[1:] - bad_idx[:-1] != 1).sum()) + 1
            print("gsx.strategy.MCMC: %d of %d Gaussians hold non-finite parameters %s (indices %d .. %d in %d contiguous runs, first %s): relocated as dead" % (
                n_bad, m.means.shape[0], per, int(bad_idx.min()), int(bad_idx.max()), runs, bad_idx[:8].tolist()), file=sys.stderr, flush=True)
            opac = torch.where(finite, opac, torch.zeros_like(opac))
            dead = dead | ~finite
            self.nonfinite_relocated = getattr(self, "nonfinite_relocated", 0) + n_bad
        dead_idx = dead.nonzero().squeeze(-1)
        n_dead = dead_idx.numel()
        if n_dead == 0:
            return 0
        alive_idx = (~dead).nonzero().squeeze(-1)
        if alive_idx.numel() == 0:
            return 0
        sampled = alive_idx.index_select(0, self._multinomial(opac.index_select(0, alive_idx), n_dead))
        ratios = torch.ones_like(opac, dtype=torch.int32)
        ratios.index_add_(0, sampled, torch.ones_like(sampled, dtype=torch.int32))
        ratios = ratios.index_select(0, sampled).clamp_max_(self.N_MAX)
        new_op, new_sc = self._relocated(opac, sampled, ratios)
        m.opacity_raw.data[sampled] = torch.logit(new_op).unsqueeze(-1)
        m.scaling_raw.data[sampled] = torch.log(new_sc)
        for t in m.params():
            t.data[dead_idx] = t.data.index_select(0, sampled)
        self.optimizer.reset_state(sampled)
        if n_bad:
            self.optimizer.reset_state(bad_idx)   # (their moments are as non-finite as they were)
        return n_dead

    # ---- mcmc.cpp:191-340 ------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def add_new_gs(self):
        m = self.model
        cur = m.means.shape[0]
        n_new = max(0, min(self.params.max_cap, int(1.05 * cur)) - cur)
        if n_new == 0:
            return 0
        opac = m.get_opacity()
        sampled = self._multinomial(opac.flatten(), n_new)
        ratios = torch.zeros(opac.shape[0], dtype=torch.float32, device=opac.device)
        ratios.index_add_(0, sampled, torch.ones_like(sampled, dtype=torch.float32))
        ratios = (ratios.index_select(0, sampled) + 1).clamp(1, self.N_MAX).to(torch.int32)
        new_op, new_sc = self._relocated(opac, sampled, ratios)
        m.opacity_raw.data[sampled] = torch.logit(new_op).unsqueeze(-1)
        m.scaling_raw.data[sampled] = torch.log(new_sc)
        for name in ("means", "sh", "scaling_raw", "rotation_raw", "opacity_raw"):
            old = getattr(m, name)
            setattr(m, name, torch.cat([old.data, old.data.index_select(0, sampled)], 0).requires_grad_(True))
        self.optimizer.extend_state(n_new)
        self._grew = True
        if self.spatial_reorder:
            self.reorder_spatially(notify=False)
        if self.on_resize is not None:
            self.on_resize(m)
        return n_new

    @torch.no_grad()
    def reorder_spatially(self, notify=True):
        """Stores the Gaussians (parameters and Adam moments alike) in Morton order of their positions (layout.py): the same model, a
        memory order under which the intersection's slices and the blend's gathers are spatially coherent.  Returns the permutation."""
        from . import layout
        if self.before_reindex is not None:
            self.before_reindex()
        order = layout.morton_order(self.model.means.data)
        for name in ("means", "sh", "scaling_raw", "rotation_raw", "opacity_raw"):
            setattr(self.model, name, getattr(self.model, name).data.index_select(0, order).requires_grad_(True))
        self.optimizer.select_state(order)
        if notify and self.on_resize is not None:
            self.on_resize(self.model)
        return order

    # ---- mcmc.cpp:342-366 ------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def inject_noise(self):
        m = self.model
        import numpy as np
        lr = float(np.float32(self.optimizer.groups[0]["lr"]) * np.float32(self.NOISE_LR))   # static_cast<float>(lr) * _noise_lr: an fp32 product (mcmc.cpp:347-349)
        noise = torch.randn(m.means.shape, dtype=m.means.dtype, device=m.means.device, generator=self.generator)
        ops.add_noise(m.opacity_raw.data.reshape(-1).contiguous(), m.scaling_raw.data, m.rotation_raw.data, noise, m.means.data, float(lr))

    # ---- mcmc.cpp:368-402 ------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def post_backward(self, it, render_output=None):
        if it % self.params.sh_degree_interval == 0:
            max_deg = int(round(self.model.sh.shape[1] ** 0.5)) - 1
            self.model.active_sh_degree = min(self.model.active_sh_degree + 1, max_deg)
        if self.is_refining(it):
            self.relocate_gs()
            self.add_new_gs()
        self.inject_noise()

    def step(self, it, optimizer_step=None):
        """mcmc.cpp:395-402 (optimizer_step: replacement for self.optimizer.step, e.g. distributed.ShardedAdam.step).  On an iteration in which add_new_gs grew the model, upstream's freshly concatenated parameter tensors
        have an undefined gradient, so FusedAdam::step skips every group (no update, no step_count++, fused_adam.cpp:33-36): the
        iteration's gradient is dropped, the scheduler still advances.  Same here."""
        if it < self.params.iterations:
            if self._grew:
                self._grew = False
            else:
                (optimizer_step or self.optimizer.step)(it)
            self.scheduler.step()

    @torch.no_grad()
    def remove_gaussians(self, mask):
        """mcmc.cpp:404-444: drop the masked Gaussians from the model and the optimizer state."""
        if int(mask.sum()) == 0:
            return
        keep = (~mask).nonzero().squeeze(-1)
        if self.before_reindex is not None:
            self.before_reindex()
        for name in ("means", "sh", "scaling_raw", "rotation_raw", "opacity_raw"):
            setattr(self.model, name, getattr(self.model, name).data.index_select(0, keep).requires_grad_(True))
        self.optimizer.select_state(keep)
        if self.on_resize is not None:
            self.on_resize(self.model)
