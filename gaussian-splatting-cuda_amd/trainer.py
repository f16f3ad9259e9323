"""Training step of the `--gut` path (the caller side of the hot path): Trainer::train_step of the reference
(src/training/trainer.cpp:579-800) restricted to what the gut / MCMC configuration executes:

    render (rasterize_fused) -> photometric loss (trainer.cpp:103-127) -> backward
    -> [N > 1 GPUs: all-reduce of the gradient rows some camera of the step saw, mean over the cameras; or (exchange="colors")
        the colour-gradient exchange: 3 floats per (camera, Gaussian) all-gathered, SH backward over all cameras on every rank]
    -> scale / opacity regularisers (trainer.cpp:132-160; their gradients are added analytically: reg * mean(exp(s)),
       reg * mean(sigmoid(o)); identical on every rank, so added after the reduction)
    -> strategy.post_backward (SH degree schedule, relocation, growth, noise) -> strategy.step (fused Adam + lr decay)

One process per GPU; rank r renders camera `cams[(it * world + r) % len(cams)]`.  Every rank holds a full replica and applies
the identical update (the strategy's random draws come from a generator seeded identically on all ranks)."""
import gc
import os
import sys

import torch
import torch.distributed as dist

from . import distributed as gdist
from . import loss as gloss
from . import rasterizer
from .strategy import MCMC, OptimizationParameters


class Trainer:
    def __init__(self, model, cameras, images, params: OptimizationParameters = None, background=None, scene_scale=1.0, seed=0,
                 sharded_adam=False, exchange="colors", fused_sh_adam=True, guarded_lists=True, fused_regularisers=True):
        """cameras: list of rasterizer.Camera; images: list of [3,H,W] ground-truth tensors on the device.
        sharded_adam (world > 1): reduce-scatter -> Adam on this rank's 1/world of the Gaussians -> all-gather of the parameters
        (distributed.ShardedAdam) instead of all-reduce + replicated Adam.
        exchange (world > 1, replicated Adam): "colors" = distributed.ColorGradExchange (default: 2.6x fewer bytes than the dense bucket, no
        host sync); "rows" = all-reduce of the gradient rows some camera of the step saw (one host sync for the row count).
        fused_sh_adam: on iterations without densification the SH tensor's Adam step is applied inside the SH backward
        (gsx_sh_colors_bwd_adam: the SH gradient is never written); needs the complete SH gradient on this rank, i.e. one GPU or the
        colour exchange.  Refine iterations keep the separate step: relocation / growth run between backward and optimizer there.
        guarded_lists: render with rasterize_fused(guarded=True) — no host read of n_isects per iteration (include/gsx.h "guarded lists");
        an iteration whose intersection lists outgrew their capacity is repeated before its optimizer step (`capacity_misses` counts them).
        fused_regularisers: the gradients of the strategy's scale / opacity regularisers are added inside the render backward's
        activation-Jacobian kernel (same values as the separate elementwise ops of _add_regularisers up to the rounding of one fma)."""
        self.model, self.cameras = model, cameras
        # dense [3,H,W] once, here: a target that is a permuted view (e.g. a clamped render: torch.clamp keeps the HWC strides of its input) would
        # otherwise be copied by the loss in EVERY iteration (13 MB per frame at 1296x840: 14 us of the garden stand-in's 2.2 ms)
        self.images = [im if im.is_contiguous() else im.contiguous() for im in images]
        self.params = params or OptimizationParameters()
        self.bg = background
        dev = model.means.device
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.multi = gdist.active()   # a gradient exchange runs (N > 1, or the one-rank diagnostic group of distributed.SINGLE_RANK_COLLECTIVES)
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed)
        for p in model.params():
            p.requires_grad_(True)
        self.exchange = exchange if (self.multi and not sharded_adam) else None
        self.fused_sh_adam = fused_sh_adam and not sharded_adam and (not self.multi or self.exchange == "colors")
        self.strategy = MCMC(model, self.params, scene_scale, gen)
        self.strategy.on_resize = self._rebuild_bucket
        self._rebuild_bucket(model)
        self.sharded = gdist.ShardedAdam(self.strategy.optimizer) if (sharded_adam and self.multi) else None
        self.guarded, self.capacity_misses = guarded_lists, 0
        self.fused_regularisers = fused_regularisers
        self._lists_agree = gdist.ListsAgreement() if (guarded_lists and self.multi) else None
        if self._lists_agree is not None:
            self.sinks["_lists_agree"] = self._lists_agree
        if self.sharded is not None:   # rows change owner when the strategy permutes / removes them: complete the moments on every rank first
            self.strategy.before_reindex = self.sharded.merge_moments
        self.last_loss = None
        self.check_finite = os.environ.get("GSX_CHECK_FINITE", "0") == "1"
        self.check_finite_async, self._finite_flags = os.environ.get("GSX_CHECK_FINITE", "0") == "2", []

    def _rebuild_bucket(self, model):
        if getattr(self, "exchange", None) == "colors":   # SH gradient first: the all-reduced remainder is one contiguous span
            names = ["sh", "means", "scaling_raw", "rotation_raw", "opacity_raw"]
            self.bucket = gdist.GradBucket([getattr(model, n) for n in names])
            self.sinks = self.bucket.sinks(tuple(names))
            self.xch = gdist.ColorGradExchange(self.bucket, names)
            self.sinks["_color_exchange"] = self.xch
        else:
            self.bucket = gdist.GradBucket(model.params())
            self.sinks = self.bucket.sinks()
        if getattr(self, "_lists_agree", None) is not None:
            self.sinks["_lists_agree"] = self._lists_agree

    def _regulariser_coefficients(self):
        """(scale_reg / numel, opacity_reg / numel): what rasterize_fused's backward adds inside the activation-Jacobian kernel
        (sinks["_regularisers"]) — the same gradients as _add_regularisers, without its six elementwise launches."""
        p, m = self.params, self.model
        return (p.scale_reg / m.scaling_raw.numel() if p.scale_reg > 0.0 else 0.0,
                p.opacity_reg / m.opacity_raw.numel() if p.opacity_reg > 0.0 else 0.0)

    @torch.no_grad()
    def _add_regularisers(self):
        if self.sinks.get("_regularisers") is not None:   # already inside the gradients (fused into the render backward)
            return
        p, m = self.params, self.model
        if p.scale_reg > 0.0:   # d/ds_raw [reg * mean(exp(s_raw))]
            m.scaling_raw.grad.add_(torch.exp(m.scaling_raw), alpha=p.scale_reg / m.scaling_raw.numel())
        if p.opacity_reg > 0.0:  # d/do_raw [reg * mean(sigmoid(o_raw))]
            s = torch.sigmoid(m.opacity_raw)
            m.opacity_raw.grad.add_(s * (1.0 - s), alpha=p.opacity_reg / m.opacity_raw.numel())

    def train_step(self, it):
        i = (it * self.world + self.rank) % len(self.cameras)
        if self.exchange == "colors":   # the step's camera batch in rank order
            self.xch.begin_step(torch.stack([self.cameras[(it * self.world + r) % len(self.cameras)].viewmat for r in range(self.world)]))
        fuse = self.fused_sh_adam and it < self.params.iterations and not self.strategy.is_refining(it)
        gt = self.images[i]
        self.sinks["_regularisers"] = self._regulariser_coefficients() if self.fused_regularisers else None
        for attempt in range(4):
            self.sinks["_sh_adam"] = self.strategy.optimizer.begin_fused_sh_step(it) if fuse else None
            fuse = self.sinks["_sh_adam"] is not None
            # guarded lists: the render never waits for n_isects; a frame whose lists outgrew their capacity stops in backward() before
            # the SH tensor's Adam step / the gradient exchange and is rendered again (the capacity hint has been raised by then)
            out = rasterizer.rasterize_fused(self.cameras[i], self.model, self.bg, grad_sinks=self.sinks, guarded=self.guarded)
            loss = gloss.photometric_loss(out.render_hwc, gt, self.params.lambda_dssim)
            try:
                gloss.backward(loss)
                break
            except rasterizer.IsectCapacityMiss as e:
                self.capacity_misses += 1
                if os.environ.get("GSX_LOG_CAPACITY_MISSES"):
                    print("iteration %d (%d Gaussians): %s" % (it, self.model.means.shape[0], e), file=sys.stderr)
                if attempt == 3:
                    raise
        if self.check_finite:
            self._check_finite(it, out, loss, "gradients")
        if self.check_finite_async:
            self._flag_finite(it, "gradients", loss)
        if self.sharded is not None:
            # the regularisers are the same on every rank, so adding them before the mean over the ranks gives the same sum; the
            # exchange itself (reduce-scatter ... all-gather) brackets the optimizer step and is skipped with it when the model grew
            self._add_regularisers()
            self.strategy.post_backward(it, out)
            self.strategy.step(it, optimizer_step=lambda i: self.sharded.step(i, self.bucket))
        else:
            if self.exchange == "colors":
                self.xch.finish()
            elif self.multi:  # only rows some camera of the step saw are non-zero: compacted all-reduce
                self.bucket.all_reduce_mean_rows((out.aux["radii_full"] > 0).all(-1))
            self._add_regularisers()  # identical on every rank (functions of the replicated parameters): added after the reduction
            self.strategy.post_backward(it, out)
            self.strategy.step(it, optimizer_step=(lambda k: self.strategy.optimizer.step(k, skip_sh=True)) if fuse else None)
        if self.check_finite:
            self._check_finite(it, out, loss, "parameters")
        if self.check_finite_async:
            self._flag_finite(it, "parameters", loss)
        self.last_loss = loss.detach()
        return self.last_loss

    _NAMES = ("means", "sh", "scaling_raw", "rotation_raw", "opacity_raw")

    def _flag_finite(self, it, what, loss):
        """GSX_CHECK_FINITE=2: the same question without a synchronisation per iteration (its reductions cost the garden stand-in's iteration a third on top):
        one flag word per tensor and iteration is left on the device and read every 250 iterations and at the end of train(); the first flagged iteration is
        reported."""
        m = self.model
        ts = [(getattr(m, n).grad if what == "gradients" else getattr(m, n)) for n in self._NAMES]
        flags = torch.stack([(~torch.isfinite(t)).any() if t is not None else torch.zeros((), dtype=torch.bool, device=m.means.device) for t in ts] +
                            [~torch.isfinite(loss.detach()).all()])
        self._finite_flags.append((it, what, m.means.shape[0], flags))
        if what == "parameters" and it % 250 == 0:
            self._read_finite_flags()

    def _read_finite_flags(self):
        """One device -> host copy of the flag words left since the last read (every 250 iterations and at the end of train()); raises on the first flagged one."""
        if self._finite_flags:
            allf = torch.stack([f for _, _, _, f in self._finite_flags]).cpu()
            if bool(allf.any()):
                k = int(allf.any(-1).nonzero()[0])
                it0, w0, n0, _ = self._finite_flags[k]
                bad = [n for n, b in zip(self._NAMES + ("loss",), allf[k].tolist()) if b]
                later = [(self._finite_flags[j][0], self._finite_flags[j][1], [n for n, b in zip(self._NAMES + ("loss",), allf[j].tolist()) if b]) for j in range(k + 1, min(k + 6, len(self._finite_flags)))]
                raise FloatingPointError("first non-finite values: iteration %d (%s, %d Gaussians), %s of %s; then %s" % (
                    it0, "refine" if self.strategy.is_refining(it0) else "plain", n0, w0, bad, later))
            self._finite_flags = []

    def _check_finite(self, it, out, loss, what):
        """GSX_CHECK_FINITE=1 (a debugging run: it synchronises every iteration): stops at the first iteration whose loss / gradients (after the backward) or
        parameters (after refine events, noise and the optimizer step) hold a non-finite value, and says where."""
        names = self._NAMES
        m = self.model
        if what == "gradients":
            bad = {n: int((~torch.isfinite(getattr(m, n).grad)).sum()) for n in names if getattr(m, n).grad is not None and n != ("sh" if self.sinks.get("_sh_adam") is not None else "")}
            bad["loss"] = 0 if bool(torch.isfinite(loss)) else 1
            bad["render"] = int((~torch.isfinite(out.render_hwc)).sum())
        else:
            bad = {n: int((~torch.isfinite(getattr(m, n))).sum()) for n in names}
        if sum(bad.values()) == 0:
            return
        rows = torch.zeros(m.means.shape[0], dtype=torch.bool, device=m.means.device)
        for n in names:
            t = getattr(m, n).grad if what == "gradients" else getattr(m, n)
            if t is not None and t.shape[0] == rows.shape[0]:
                rows |= ~torch.isfinite(t.reshape(t.shape[0], -1)).all(-1)
        idx = rows.nonzero().flatten()
        k = idx[:6]
        sc, op = m.get_scaling(), m.get_opacity().flatten()
        msg = ["iteration %d (%s, %d Gaussians): non-finite %s: %s; %d Gaussians affected" % (it, "refine" if self.strategy.is_refining(it) else "plain", m.means.shape[0], what, bad, idx.numel())]
        if k.numel():
            msg.append("  e.g. Gaussians %s: opacity %s scales %s |rotation_raw| %s radii %s" % (k.tolist(), op[k].tolist(), sc[k].tolist(), m.rotation_raw[k].norm(dim=-1).tolist(),
                                                                                              out.aux["radii_full"].reshape(-1, 2)[k].tolist() if out is not None and k.max() < out.aux["radii_full"].numel() // 2 else "-"))
        fin = torch.isfinite(sc).all(-1)
        msg.append("  model: scales %.3g .. %.3g (ratio max %.3g), opacity %.3g .. %.3g, n_isects %s" % (float(sc[fin].min()), float(sc[fin].max()), float((sc[fin].max(-1).values / sc[fin].min(-1).values).max()),
                                                                                                     float(op[torch.isfinite(op)].min()), float(op[torch.isfinite(op)].max()), getattr(out, "n_isects", "-")))
        raise FloatingPointError("\n".join(msg))

    def train(self, iterations=None, start=1, log_every=0):
        n = iterations or self.params.iterations
        # the objects alive now (modules, the dataset's tensors) move to the collector's permanent generation: the collections the loop still
        # triggers stay short — a full one takes milliseconds, and the host is at most one iteration (1 - 2 ms) ahead of the GPU
        gc.collect()
        gc.freeze()
        try:
            for it in range(start, start + n):
                loss = self.train_step(it)
                if log_every and it % log_every == 0 and self.rank == 0:
                    done = it - start + 1
                    print(f"iter {it}: loss {float(loss):.5f}  gaussians {self.model.means.shape[0]}  repeated iterations (list capacity misses) "
                          f"{self.capacity_misses} = {1e3 * self.capacity_misses / done:.2f} per 1000")
        finally:
            gc.unfreeze()
        if self.check_finite_async:
            self._read_finite_flags()   # (a run that does not end on a multiple of 250)
        return self.last_loss
