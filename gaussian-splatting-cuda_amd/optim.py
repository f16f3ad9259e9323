"""Optimizer of the training loop: the reference's FusedAdam and ExponentialLR
(src/training/optimizers/fused_adam.cpp:20-96, scheduler.cpp:10-25, group set-up strategy_utils.cpp:20-55) on the
fused HIP Adam kernel (csrc/gsx_adam.hip).

Parameter groups, in the reference's order: means, sh0, shN, scaling, rotation, opacity.  Here sh0 / shN are the two
column blocks of ONE [N,K,3] SH tensor (rasterizer.SplatData); each block keeps its own learning rate and Adam state and
is updated by one strided launch.  Quirks kept from the reference (fused_adam.cpp:68-76): the shN group (i == 3) is not
stepped during the first 1000 iterations although its step counter advances; with `skip_sh_steps` it is stepped only
every second iteration until iteration 25000."""
import math

import torch

from . import ops


class FusedAdam:
    def __init__(self, groups, betas=(0.9, 0.999), eps=1e-8, skip_sh_steps=False):
        """groups: list of dicts {"name", "param" (leaf tensor or view of one), "grad" (callable -> tensor or None), "lr"}."""
        self.groups = groups
        self.betas, self.eps, self.skip_sh_steps = betas, eps, skip_sh_steps
        self.state = {}

    @staticmethod
    def for_splat_data(model, means_lr=0.00016, shs_lr=0.0025, scaling_lr=0.005, rotation_lr=0.001, opacity_lr=0.05,
                       scene_scale=1.0, **kw):
        """strategy_utils.cpp:35-40 (default_optimization_params: parameters.hpp:19-23)."""
        def g(name, param, grad, lr):
            return {"name": name, "param": param, "grad": grad, "lr": lr}
        sh_grad = lambda: model.sh.grad  # noqa: E731
        return FusedAdam([
            g("means", model.means, lambda: model.means.grad, means_lr * scene_scale),
            g("sh0", model.sh.data[:, :1], lambda: None if sh_grad() is None else sh_grad()[:, :1], shs_lr),
            g("shN", model.sh.data[:, 1:], lambda: None if sh_grad() is None else sh_grad()[:, 1:], shs_lr / 20.0),
            g("scaling", model.scaling_raw, lambda: model.scaling_raw.grad, scaling_lr),
            g("rotation", model.rotation_raw, lambda: model.rotation_raw.grad, rotation_lr),
            g("opacity", model.opacity_raw, lambda: model.opacity_raw.grad, opacity_lr),
        ], **kw)

    @torch.no_grad()
    def step(self, iteration):
        b1, b2 = self.betas
        for i, grp in enumerate(self.groups, start=1):
            grad = grp["grad"]()
            if grad is None:
                continue
            st = self.state.get(grp["name"])
            if st is None:
                p = grp["param"]
                st = self.state[grp["name"]] = {"step": 0, "exp_avg": torch.zeros(p.shape, dtype=p.dtype, device=p.device),
                                                "exp_avg_sq": torch.zeros(p.shape, dtype=p.dtype, device=p.device)}
            st["step"] += 1
            if i == 3 and iteration <= 1000:
                continue
            if self.skip_sh_steps and i == 3 and (iteration % 2 != 0 and iteration <= 25000):
                continue
            bc1_rcp = 1.0 / (1.0 - math.pow(b1, st["step"]))
            bc2_sqrt_rcp = 1.0 / math.sqrt(1.0 - math.pow(b2, st["step"]))
            p = grp["param"].data if isinstance(grp["param"], torch.Tensor) else grp["param"]
            ops.adam_step(p, st["exp_avg"], st["exp_avg_sq"], grad, grp["lr"], b1, b2, self.eps, bc1_rcp, bc2_sqrt_rcp)


class ExponentialLR:
    """scheduler.cpp:10-25; gamma = 0.01 ** (1 / iterations) on the means group (strategy_utils.cpp:50-54)."""

    def __init__(self, optimizer, gamma, param_group_index=0):
        self.optimizer, self.gamma, self.idx = optimizer, gamma, param_group_index

    def step(self):
        groups = self.optimizer.groups if self.idx < 0 else [self.optimizer.groups[self.idx]]
        for g in groups:
            g["lr"] *= self.gamma
