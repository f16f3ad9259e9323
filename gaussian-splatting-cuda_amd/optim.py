"""Optimizer of the training loop: the reference's FusedAdam and ExponentialLR
(src/training/optimizers/fused_adam.cpp:20-96, scheduler.cpp:10-25, group set-up strategy_utils.cpp:20-55 /
strategies/mcmc.cpp:476-499) on the fused HIP Adam kernels (csrc/gsx_adam.hip).

Parameter groups, in the reference's order: means, sh0, shN, scaling, rotation, opacity; betas (0.9, 0.999), eps 1e-15.
Here sh0 / shN are the two column blocks of ONE [N,K,3] SH tensor (rasterizer.SplatData); each block keeps its own
learning rate and is updated, together with the other block, by one vectorised launch over the dense tensor
(gsx_adam_step_split; the two groups always share their step count, fused_adam.cpp:66).  Quirks kept from the reference
(fused_adam.cpp:68-76): the shN group (i == 3) is not stepped during the first 1000 iterations although its step counter
advances; with `skip_sh_steps` (a compile-time switch upstream, default off) it is stepped only every second iteration
until iteration 25000.

Groups address the model's tensors by name at step time, so a densification strategy may replace them by longer ones
(`extend_state`) or reset the moments of relocated Gaussians (`reset_state`), as strategies/mcmc.cpp:85-112,252-330 do."""
import math

import torch

from . import ops

GROUPS = ("means", "sh0", "shN", "scaling", "rotation", "opacity")
_ATTR = {"means": "means", "sh0": "sh", "shN": "sh", "scaling": "scaling_raw", "rotation": "rotation_raw", "opacity": "opacity_raw"}


class FusedAdam:
    def __init__(self, model, lrs, betas=(0.9, 0.999), eps=1e-15, skip_sh_steps=False):
        """lrs: dict group name -> learning rate (all six of GROUPS)."""
        self.model = model
        self.groups = [{"name": n, "lr": float(lrs[n])} for n in GROUPS]
        self.betas, self.eps, self.skip_sh_steps = betas, eps, skip_sh_steps
        self.state = {}   # "means" | "sh" | "scaling" | "rotation" | "opacity" -> exp_avg / exp_avg_sq; step counts per group
        self.reindex_generation = 0   # bumped whenever rows are permuted / removed (select_state): distributed.ShardedAdam checks it

    @staticmethod
    def for_splat_data(model, means_lr=0.00016, shs_lr=0.0025, scaling_lr=0.005, rotation_lr=0.001, opacity_lr=0.05,
                       scene_scale=1.0, **kw):
        """strategy_utils.cpp:35-40 (defaults: include/core/parameters.hpp:19-23).  The reference's parameters are C floats and the group
        learning rates are formed in fp32 before they are widened to the optimizer's double (`_params->means_lr * get_scene_scale()`,
        `_params->shs_lr / 20.f`): the same here — 0.00016f is 0.00015999999595806003, not 0.00016, and the scheduler multiplies that double every
        iteration (found by running the reference's own mcmc.cpp next to this class: tests/test_gpu_reference_strategy.py)."""
        import numpy as np
        f = np.float32
        return FusedAdam(model, {"means": float(f(means_lr) * f(scene_scale)), "sh0": float(f(shs_lr)), "shN": float(f(shs_lr) / f(20.0)),
                                 "scaling": float(f(scaling_lr)), "rotation": float(f(rotation_lr)), "opacity": float(f(opacity_lr))}, **kw)

    # ---- state ---------------------------------------------------------------------------------------------------------
    def _param(self, name):
        return getattr(self.model, _ATTR[name])

    def _moments(self, name):
        """Moments of a group; sh0 / shN share one [N,K,3] pair when the split launch applies, else one dense pair per block."""
        sh = self.model.sh
        split = (sh.shape[1] * 3) % 4 == 0 and sh.shape[1] > 1
        key = _ATTR[name] if (split or name not in ("sh0", "shN")) else name
        st = self.state.get(key)
        if st is None:
            p = self._param(name)
            if key == "sh0":
                p = p[:, :1]
            elif key == "shN":
                p = p[:, 1:]
            st = self.state[key] = {"exp_avg": torch.zeros(p.shape, dtype=p.dtype, device=p.device),
                                    "exp_avg_sq": torch.zeros(p.shape, dtype=p.dtype, device=p.device)}
        return st

    def step_count(self, name):
        return self.state.get("step:" + name, 0)

    def reset_state(self, indices):
        """Zero both moments of the given Gaussians in every group (MCMC::update_optimizer_for_relocate, mcmc.cpp:85-112)."""
        for key, st in self.state.items():
            if isinstance(st, dict):
                st["exp_avg"][indices] = 0
                st["exp_avg_sq"][indices] = 0

    def extend_state(self, n_new):
        """The model grew by n_new Gaussians at the end: append zero moments, keep the step counts (mcmc.cpp:264-318)."""
        for key, st in self.state.items():
            if isinstance(st, dict):
                for k in ("exp_avg", "exp_avg_sq"):
                    z = torch.zeros((n_new,) + tuple(st[k].shape[1:]), dtype=st[k].dtype, device=st[k].device)
                    st[k] = torch.cat([st[k], z], 0)

    def select_state(self, indices):
        """Keep only the given Gaussians (MCMC::remove_gaussians, mcmc.cpp:404-444) — or permute them (MCMC.reorder_spatially)."""
        self.reindex_generation += 1
        for key, st in self.state.items():
            if isinstance(st, dict):
                st["exp_avg"] = st["exp_avg"].index_select(0, indices)
                st["exp_avg_sq"] = st["exp_avg_sq"].index_select(0, indices)

    # ---- step ------------------------------------------------------------------------------------------------------------
    def begin_fused_sh_step(self, iteration):
        """The SH groups' share of step(iteration), handed to the render backward: the fused kernel gsx_sh_colors_bwd_adam applies it
        where the SH gradient is produced (the 192 MB gradient is then never written or re-read).  Computes the sh0 / shN step counters
        exactly as step() would (they are committed by the following step(skip_sh=True)) and returns the kernel's arguments; the following step(iteration, skip_sh=True) updates the other groups.
        Returns None when the split launch does not apply (K * 3 not a multiple of 4): call step() as usual then."""
        sh = self.model.sh
        self._pending_sh = None
        # (the kernel updates the tensor the render saved, `sh.contiguous()`: a non-contiguous model.sh would be stepped on a temporary copy)
        if not ((sh.shape[1] * 3) % 4 == 0 and sh.shape[1] > 1 and sh.is_contiguous()):
            return None
        b1, b2 = self.betas
        args, pending = {}, {}
        for i, grp in enumerate(self.groups, start=1):
            name = grp["name"]
            if name not in ("sh0", "shN"):
                continue
            pending[name] = t = self.step_count(name) + 1   # committed by step(skip_sh=True), i.e. after the fused kernel ran
            skip = i == 3 and (iteration <= 1000 or (self.skip_sh_steps and iteration % 2 != 0 and iteration <= 25000))
            args[name] = (not skip, grp["lr"] / (1.0 - math.pow(b1, t)), 1.0 / math.sqrt(1.0 - math.pow(b2, t)))
        assert args["sh0"][2] == args["shN"][2], "sh0 / shN step counters diverged"
        self._pending_sh = pending
        st = self._moments("sh0")
        # positional tail of ops.sh_colors_bwd_adam: exp_avg, exp_avg_sq, step_sh0, step_shN, do_sh0, do_shN, beta1, beta2, eps, bc2_sqrt_rcp
        return (st["exp_avg"], st["exp_avg_sq"], args["sh0"][1], args["shN"][1], args["sh0"][0], args["shN"][0], b1, b2, self.eps, args["sh0"][2])

    @torch.no_grad()
    def step(self, iteration, rows=None, skip_sh=False):
        """One Adam step of every group.  rows=(lo, hi) restricts the update to that block of Gaussians (distributed.ShardedAdam:
        the other blocks are updated by the other ranks); step counters and bias corrections advance as for a full step.
        skip_sh: the sh0 / shN groups were stepped by the fused render backward (begin_fused_sh_step)."""
        b1, b2 = self.betas
        sh = self.model.sh
        sh_grad = sh.grad
        if skip_sh:   # the fused render backward stepped the SH tensor: only now do its step counters advance
            pending = getattr(self, "_pending_sh", None)
            if not pending:
                raise RuntimeError("step(skip_sh=True) without a begin_fused_sh_step() that returned kernel arguments")
            for name, t in pending.items():
                self.state["step:" + name] = t
        self._pending_sh = None
        r = (lambda t: t) if rows is None else (lambda t: t[rows[0]:rows[1]])
        split_ok = (sh.shape[1] * 3) % 4 == 0 and sh.shape[1] > 1
        do_sh, dense = {}, []
        for i, grp in enumerate(self.groups, start=1):
            name = grp["name"]
            p = self._param(name)
            grad = sh_grad if name in ("sh0", "shN") else p.grad
            if grad is None or (name == "shN" and sh.shape[1] == 1) or (skip_sh and name in ("sh0", "shN")):
                continue
            self.state["step:" + name] = t = self.step_count(name) + 1
            skip = i == 3 and (iteration <= 1000 or (self.skip_sh_steps and iteration % 2 != 0 and iteration <= 25000))
            bc1_rcp = 1.0 / (1.0 - math.pow(b1, t))
            bc2_sqrt_rcp = 1.0 / math.sqrt(1.0 - math.pow(b2, t))
            if name in ("sh0", "shN"):
                do_sh[name] = (not skip, grp["lr"], bc1_rcp, bc2_sqrt_rcp)
                continue
            if skip:
                continue
            st = self._moments(name)
            dense.append((r(p.data), r(st["exp_avg"]), r(st["exp_avg_sq"]), r(grad), grp["lr"], bc1_rcp, bc2_sqrt_rcp))
        if dense and all(t.is_contiguous() for d in dense for t in d[:4]):
            # the dense groups (means, scaling, rotation, opacity) in ONE launch: four launches of 4-16 MB each leave the chip half idle
            cols = list(zip(*dense))
            ops.adam_step_multi(list(cols[0]), list(cols[1]), list(cols[2]), list(cols[3]), list(cols[4]), list(cols[5]), list(cols[6]), b1, b2, self.eps)
        else:
            for d in dense:
                ops.adam_step(d[0], d[1], d[2], d[3], d[4], b1, b2, self.eps, d[5], d[6])
        if do_sh:
            a = do_sh.get("sh0", (False, 0.0, 1.0, 1.0))
            b = do_sh.get("shN", (False, 0.0, a[2], a[3]))
            if split_ok and a[2:] == b[2:] and sh_grad.is_contiguous():
                st = self._moments("sh0")
                ops.adam_step_split(r(sh.data), r(st["exp_avg"]), r(st["exp_avg_sq"]), r(sh_grad), 3, a[1], b[1], a[0], b[0], b1, b2, self.eps, a[2], a[3])
            else:  # K = 1 / 9 / 25 (K*3 not a multiple of 4): one row-strided launch per block, dense moments per block
                assert not split_ok, "sh0 / shN step counters diverged"
                if a[0]:
                    st = self._moments("sh0")
                    ops.adam_step(r(sh.data)[:, :1], r(st["exp_avg"]), r(st["exp_avg_sq"]), r(sh_grad)[:, :1], a[1], b1, b2, self.eps, a[2], a[3])
                if b[0]:
                    st = self._moments("shN")
                    ops.adam_step(r(sh.data)[:, 1:], r(st["exp_avg"]), r(st["exp_avg_sq"]), r(sh_grad)[:, 1:], b[1], b1, b2, self.eps, b[2], b[3])

    def zero_grad(self, set_to_none=True):
        for p in self.model.params():
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.zero_()


class ExponentialLR:
    """scheduler.cpp:10-25; gamma = 0.01 ** (1 / iterations) on the means group (strategy_utils.cpp:50-54)."""

    def __init__(self, optimizer, gamma, param_group_index=0):
        self.optimizer, self.gamma, self.idx = optimizer, gamma, param_group_index

    def step(self):
        groups = self.optimizer.groups if self.idx < 0 else [self.optimizer.groups[self.idx]]
        for g in groups:
            g["lr"] *= self.gamma
