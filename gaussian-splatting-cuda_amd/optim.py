"""Optimizer of the training loop: the reference's FusedAdam and ExponentialLR
(src/training/optimizers/fused_adam.cpp:20-96, scheduler.cpp:10-25, group set-up strategy_utils.cpp:20-55) on the
fused HIP Adam kernel (csrc/gsx_adam.hip).

Parameter groups, in the reference's order: means, sh0, shN, scaling, rotation, opacity.  Here sh0 / shN are the two
column blocks of ONE [N,K,3] SH tensor (rasterizer.SplatData); each block keeps its own learning rate and Adam state and
is updated by one vectorised launch over the dense tensor (gsx_adam_step_split; the two groups always share their step
count, fused_adam.cpp:66).  Quirks kept from the reference (fused_adam.cpp:68-76): the shN group (i == 3) is not
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
        sh0 = g("sh0", model.sh.data[:, :1], lambda: None if sh_grad() is None else sh_grad()[:, :1], shs_lr)
        shN = g("shN", model.sh.data[:, 1:], lambda: None if sh_grad() is None else sh_grad()[:, 1:], shs_lr / 20.0)
        if (model.sh.shape[1] * 3) % 4 == 0:  # K = 4 / 16: one vectorised launch over the dense tensor; else two strided ones
            sh0.update(parent=model.sh, parent_grad=sh_grad)
            shN.update(parent=model.sh, parent_grad=sh_grad)
        return FusedAdam([
            g("means", model.means, lambda: model.means.grad, means_lr * scene_scale),
            sh0, shN,
            g("scaling", model.scaling_raw, lambda: model.scaling_raw.grad, scaling_lr),
            g("rotation", model.rotation_raw, lambda: model.rotation_raw.grad, rotation_lr),
            g("opacity", model.opacity_raw, lambda: model.opacity_raw.grad, opacity_lr),
        ], **kw)

    @torch.no_grad()
    def step(self, iteration):
        b1, b2 = self.betas
        pending = None  # the sh0 group waiting for shN: both blocks of one dense SH tensor go out in one launch
        for i, grp in enumerate(self.groups, start=1):
            grad = grp["grad"]()
            if grad is None:
                continue
            st = self.state.get(grp["name"])
            if st is None:
                p = grp["param"]
                st = self.state[grp["name"]] = {"step": 0}
                if "parent" not in grp or grp["name"] == "sh0":
                    shape = grp["parent"].shape if "parent" in grp else p.shape
                    st["exp_avg"] = torch.zeros(shape, dtype=p.dtype, device=p.device)
                    st["exp_avg_sq"] = torch.zeros(shape, dtype=p.dtype, device=p.device)
            st["step"] += 1
            skip = i == 3 and (iteration <= 1000 or (self.skip_sh_steps and iteration % 2 != 0 and iteration <= 25000))
            if "parent" in grp:                       # sh0 / shN blocks of the single SH tensor
                if grp["name"] == "sh0":
                    pending = (grp, st)
                    continue
                g0, st0 = pending
                pending = None
                assert st0["step"] == st["step"], "sh0 / shN step counters diverged"
                bc1_rcp = 1.0 / (1.0 - math.pow(b1, st["step"]))
                bc2_sqrt_rcp = 1.0 / math.sqrt(1.0 - math.pow(b2, st["step"]))
                parent = grp["parent"]
                ops.adam_step_split(parent.data, st0["exp_avg"], st0["exp_avg_sq"], grp["parent_grad"](), 3, g0["lr"], grp["lr"], True, not skip,
                                    b1, b2, self.eps, bc1_rcp, bc2_sqrt_rcp)
                continue
            if skip:
                continue
            bc1_rcp = 1.0 / (1.0 - math.pow(b1, st["step"]))
            bc2_sqrt_rcp = 1.0 / math.sqrt(1.0 - math.pow(b2, st["step"]))
            p = grp["param"].data if isinstance(grp["param"], torch.Tensor) else grp["param"]
            ops.adam_step(p, st["exp_avg"], st["exp_avg_sq"], grad, grp["lr"], b1, b2, self.eps, bc1_rcp, bc2_sqrt_rcp)
        if pending is not None:                       # sh0 had a gradient but shN did not: step the block alone
            g0, st0 = pending
            ops.adam_step_split(g0["parent"].data, st0["exp_avg"], st0["exp_avg_sq"], g0["parent_grad"](), 3, g0["lr"], 0.0, True, False, b1, b2,
                                self.eps, 1.0 / (1.0 - math.pow(b1, st0["step"])), 1.0 / math.sqrt(1.0 - math.pow(b2, st0["step"])))


class ExponentialLR:
    """scheduler.cpp:10-25; gamma = 0.01 ** (1 / iterations) on the means group (strategy_utils.cpp:50-54)."""

    def __init__(self, optimizer, gamma, param_group_index=0):
        self.optimizer, self.gamma, self.idx = optimizer, gamma, param_group_index

    def step(self):
        groups = self.optimizer.groups if self.idx < 0 else [self.optimizer.groups[self.idx]]
        for g in groups:
            g["lr"] *= self.gamma
