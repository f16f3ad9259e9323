"""Camera-sharded data parallelism for the hot path (new functionality: the reference is single-GPU,
SURVEY.md §0.4 / §8e).  One process per GPU, full parameter replica per rank, rank r renders camera r of the
step's batch; ONE collective per step: an all-reduce (sum, then 1/world) of the per-Gaussian gradients.

All six parameter gradients live in ONE flat fp32 bucket (59 floats per Gaussian at SH degree 3: means 3,
sh0 3, shN 45, scaling 3, rotation 4, opacity 1) allocated once; `.grad` of every parameter is a view into it,
so autograd accumulates straight into the bucket and the step issues a single large all-reduce — on MI355X's
point-to-point xGMI mesh one big message lets RCCL spread the reduce-scatter/all-gather over all 7 links."""
import contextlib
import os
import sys

import torch
import torch.distributed as dist


# A process group of ONE rank skips every collective (nothing to exchange).  True = run them all the same: the whole N > 1 code path —
# bucket all-reduces, colour-gradient all-gather, in-place reduce-scatter / all-gather of ShardedAdam, the gloo side group — through the
# real backend on a 1-GPU box (RCCL's argument checks, stream hand-over, in-place aliasing rules; results must equal the plain step's).
# Set by init_from_env when GSX_SINGLE_RANK_GROUP=1 (tests/test_gpu_distributed.py, `GSX_SINGLE_RANK_GROUP=1 python bench.py`).
SINGLE_RANK_COLLECTIVES = False


@contextlib.contextmanager
def _stdout_to_stderr():
    """gloo reports its mesh ("[Gloo] Rank 0 is connected to 7 peer ranks. ...") on STDOUT, from C++, whenever a gloo group is created.  A
    launcher that reads rank 0's stdout as the program's result (bench.py: ONE JSON line) must not find that there: while a group is being
    created, file descriptor 1 points at stderr."""
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def active():
    """True when the step has a gradient exchange to run."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or SINGLE_RANK_COLLECTIVES)


def init_from_env(backend=None):
    """torch.distributed.run contract: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment."""
    global SINGLE_RANK_COLLECTIVES
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and os.environ.get("GSX_SINGLE_RANK_GROUP") == "1" and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:   # a free port of this host: two single-rank runs side by side must not meet on a fixed one
            import socket
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(so.getsockname()[1])
        SINGLE_RANK_COLLECTIVES = True
    if (world > 1 or SINGLE_RANK_COLLECTIVES) and not dist.is_initialized():
        # this host's driver only supports dmabuf IPC: without it RCCL's peer mappings fail with `hipIpcGetMemHandle: invalid argument`.
        # The HSA runtime reads the variable when it starts, i.e. with the process's first GPU call: a value the launcher exported wins
        # (bench.py's self-launch and the driver's environment do); setting it here only helps a process that has not touched the GPU yet.
        if "HSA_ENABLE_IPC_MODE_LEGACY" not in os.environ:
            if world > 1 and torch.cuda.is_available() and torch.cuda.is_initialized():
                print("gsx.distributed: HSA_ENABLE_IPC_MODE_LEGACY is not set and this process has already initialised the GPU runtime; export "
                      "HSA_ENABLE_IPC_MODE_LEGACY=0 before the first GPU call or RCCL's peer mappings may fail (hipIpcGetMemHandle: invalid argument)",
                      file=sys.stderr)
            os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"  # "nccl" IS RCCL on ROCm
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        with _stdout_to_stderr():
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


class GradBucket:
    """Flat gradient bucket over a list of parameters; .grad of each parameter is a view into it."""

    def __init__(self, params):
        self.params = list(params)
        align = 64  # elements: every view starts on a 256 B boundary (the vectorised kernels need 16 B; pads are reduced along)
        offs, off = [], 0
        for p in self.params:
            offs.append(off)
            off += (p.numel() + align - 1) // align * align
        p0 = self.params[0]
        self.flat = torch.zeros(off, dtype=p0.dtype, device=p0.device)
        self.offsets = offs
        for p, o in zip(self.params, offs):
            p.grad = self.flat[o:o + p.numel()].view_as(p)

    def sinks(self, names=("means", "sh", "scaling_raw", "rotation_raw", "opacity_raw")):
        """name -> gradient view, for rasterize_fused(grad_sinks=...): backward overwrites the bucket in place."""
        assert len(names) == len(self.params)
        return {n: p.grad for n, p in zip(names, self.params)}

    def zero_(self):
        self.flat.zero_()

    _avg_ok = None  # does the backend average inside the collective?  Probed once, identically on every rank.

    @classmethod
    def _reduce_mean(cls, t):
        """In-place mean over the ranks.  RCCL averages inside the collective (no extra pass over the bucket); gloo sums."""
        if cls._avg_ok is None:
            cls._avg_ok = False
            if dist.get_backend() == "nccl":
                try:
                    probe = torch.ones(1, dtype=t.dtype, device=t.device)
                    dist.all_reduce(probe, op=dist.ReduceOp.AVG)
                    cls._avg_ok = True
                except Exception:  # noqa: BLE001  (an RCCL build without ncclAvg: every rank lands here)
                    cls._avg_ok = False
        if cls._avg_ok:
            dist.all_reduce(t, op=dist.ReduceOp.AVG)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            t.mul_(1.0 / dist.get_world_size())

    def all_reduce_mean(self, async_op=False):
        """Mean of the bucket over the ranks.  async_op=True returns a handle whose wait() completes the MEAN (the division is part
        of the collective on RCCL, applied in wait() otherwise), so work that does not read the gradients can be enqueued meanwhile."""
        if not active():
            return None
        self.last_reduced_bytes = self.nbytes()
        if async_op:
            return _MeanHandle(self.flat)
        self._reduce_mean(self.flat)
        return None

    # ---- overlapped exchange: the gradients of the late parameters first ------------------------------------------------------------
    def all_reduce_mean_tail_async(self, first_param):
        """Starts the mean all-reduce of the bucket from parameter `first_param` to the end and returns a handle (wait() completes it).
        The render backward produces the scaling / rotation / opacity gradients (the last three parameters: 8 of the 59 floats per
        Gaussian) BEFORE the SH backward, so their exchange runs under it: rasterize_fused(grad_sinks={..., "_early_ready": callback})."""
        if not active():
            return None
        self._tail_from = self.offsets[first_param]
        return _MeanHandle(self.flat[self._tail_from:])

    def all_reduce_mean_head(self, tail_handle):
        """Completes the exchange started by all_reduce_mean_tail_async: reduces the head of the bucket, waits for the tail."""
        if tail_handle is None:
            return self.all_reduce_mean()
        self.last_reduced_bytes = self.nbytes()
        self._reduce_mean(self.flat[:self._tail_from])
        tail_handle.wait()
        return None

    def all_reduce_mean_rows(self, visible, dense_above=0.75):
        """Same result as all_reduce_mean() when the gradient rows of Gaussians outside `visible` ([N] bool / uint8, this rank's
        camera) are zero — true for the render backward: a Gaussian no camera of the step sees has a zero gradient row on every
        rank.  Only the union of the visible rows travels: one small MAX all-reduce of the mask, then ONE all-reduce of the
        compacted rows (at S-1M a camera sees ~40 % of the Gaussians: 94 MB instead of 236 MB over xGMI, where the ring is
        per-link bound).  Falls back to the dense collective when the union exceeds `dense_above` of the rows."""
        if not active():
            return None
        n = self.params[0].shape[0]
        assert all(p.shape[0] == n for p in self.params), "row compaction needs per-Gaussian parameters"
        # When the last probe found (almost) everything visible, the mask exchange and its host sync are pure overhead: go dense
        # for the next 32 steps, then probe again.  Every rank sees the same union, so every rank takes the same branch.
        if getattr(self, "_dense_steps_left", 0) > 0:
            self._dense_steps_left -= 1
            return self.all_reduce_mean()
        mask = visible.reshape(-1).to(torch.uint8).contiguous()
        dist.all_reduce(mask, op=dist.ReduceOp.MAX)
        idx = mask.nonzero().squeeze(1)              # one host sync: every rank learns the same row count
        m = idx.numel()
        if m > dense_above * n:
            self._dense_steps_left = 32
            self.last_reduced_bytes = self.nbytes()
            self._reduce_mean(self.flat)
            return None
        self.last_reduced_bytes = mask.numel() + m * (self.flat.element_size() * sum(p.numel() // n for p in self.params))
        if m == 0:
            return None
        widths = [p.numel() // n for p in self.params]
        need = m * sum(widths)
        if getattr(self, "_compact", None) is None or self._compact.numel() < need:
            self._compact = torch.empty(int(need * 1.25) + 64, dtype=self.flat.dtype, device=self.flat.device)
        views, off = [], 0
        for p, w in zip(self.params, widths):
            v = self._compact[off:off + m * w].view((m,) + tuple(p.shape[1:]))
            torch.index_select(p.grad, 0, idx, out=v)
            views.append(v)
            off += m * w
        self._reduce_mean(self._compact[:off])
        for p, v in zip(self.params, views):
            p.grad.index_copy_(0, idx, v)
        return None

    def nbytes(self):
        return self.flat.numel() * self.flat.element_size()


class _MeanHandle:
    """Asynchronous mean all-reduce of a tensor: wait() returns once the tensor holds the mean on the current stream."""

    def __init__(self, t):
        self.t = t
        if GradBucket._avg_ok is None:  # probe once (synchronously, on a 1-element tensor)
            GradBucket._reduce_mean(torch.zeros(1, dtype=t.dtype, device=t.device))
        self.scale = None if GradBucket._avg_ok else 1.0 / dist.get_world_size()
        self.work = dist.all_reduce(t, op=dist.ReduceOp.AVG if GradBucket._avg_ok else dist.ReduceOp.SUM, async_op=True)

    def wait(self):
        self.work.wait()
        if self.scale is not None:
            self.t.mul_(self.scale)
            self.scale = None


class ShardedAdam:
    """Gradient exchange + optimizer step of the camera-sharded step as reduce-scatter -> Adam on 1/world of the Gaussians ->
    all-gather of the updated parameters.  Same bytes over xGMI as the all-reduce (a ring all-reduce IS a reduce-scatter followed by
    an all-gather), but the optimizer touches 1/world of the rows on every rank (59 M parameters: 0.27 ms -> 0.03 ms at 8 GPUs) and
    the second half of the exchange carries parameters instead of gradients.  Every rank ends the step with bit-identical
    parameters (the all-gather distributes ONE computed copy of each row).

    Rows are split in `world` contiguous blocks of ceil(n / world).  When n is divisible by the world size the exchange is RCCL's
    in-place reduce-scatter / all-gather; otherwise (while the model grows) the gradient is all-reduced and the updated blocks are
    merged with one SUM all-reduce of the parameters with the foreign rows zeroed (each row has exactly one owner: same bits).
    The Adam moments stay full-size on every rank (the densification strategy indexes them by global row) but only the owner's copy
    of a row is current; when the partition moves (the Gaussian count changed) the moments are merged the same way first."""

    def __init__(self, optimizer, bucket=None):
        self.opt, self.bucket = optimizer, bucket
        self._bounds = None   # (n, lo, hi) of the partition the moments were last updated under
        self._gen = getattr(optimizer, "reindex_generation", 0)   # the optimizer's row order those bounds refer to

    @staticmethod
    def _merge_owned_rows(t, lo, hi):
        """Every rank ends with the owners' rows: zero the rows this rank does not own, SUM over the ranks."""
        t[:lo].zero_()
        t[hi:].zero_()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)

    def _reduce_mean(self, g, lo, hi, even):
        world = dist.get_world_size()
        if even and dist.get_backend() == "nccl":  # RCCL: in-place reduce-scatter (output = this rank's block of the input)
            if GradBucket._avg_ok is None:
                GradBucket._reduce_mean(torch.zeros(1, dtype=g.dtype, device=g.device))
            flat = g.reshape(-1)
            per = flat.numel() // world
            out = flat[dist.get_rank() * per:(dist.get_rank() + 1) * per]
            dist.reduce_scatter_tensor(out, flat, op=dist.ReduceOp.AVG if GradBucket._avg_ok else dist.ReduceOp.SUM)
            if not GradBucket._avg_ok:
                out.mul_(1.0 / world)
        else:  # gloo has no reduce-scatter; uneven blocks: all-reduce, every rank then reads its block only
            dist.all_reduce(g, op=dist.ReduceOp.SUM)
            g[lo:hi].mul_(1.0 / world)

    def _gather_rows(self, p, lo, hi, even):
        if even and dist.get_backend() == "nccl":  # RCCL: in-place all-gather (input = this rank's block of the output)
            flat = p.reshape(-1)
            per = flat.numel() // dist.get_world_size()
            dist.all_gather_into_tensor(flat, flat[dist.get_rank() * per:(dist.get_rank() + 1) * per])
        else:
            self._merge_owned_rows(p, lo, hi)

    @torch.no_grad()
    def merge_moments(self):
        """Bring every rank's Adam moments up to date under the current partition (each rank only stepped its own rows).  To be called
        BEFORE the model is re-indexed or shrunk; afterwards the next step() starts a fresh partition."""
        if self._bounds is None:
            return
        n_old, lo_old, hi_old = self._bounds
        for st in self.opt.state.values():
            if isinstance(st, dict):
                for k in ("exp_avg", "exp_avg_sq"):
                    if st[k].shape[0] >= n_old:
                        self._merge_owned_rows(st[k][:n_old], lo_old, hi_old)
        self._bounds = None

    @torch.no_grad()
    def step(self, iteration, bucket=None):
        world, rank = dist.get_world_size(), dist.get_rank()
        self.bucket = bucket if bucket is not None else self.bucket
        params = self.bucket.params
        n = params[0].shape[0]
        assert all(p.shape[0] == n for p in params), "row sharding needs per-Gaussian parameters"
        per = (n + world - 1) // world
        lo, hi = min(n, rank * per), min(n, (rank + 1) * per)
        even = n % world == 0
        gen = getattr(self.opt, "reindex_generation", 0)
        if self._bounds is not None and gen != self._gen:
            # rows were permuted or removed (FusedAdam.select_state) while this rank held current moments only for ITS rows: a same-size
            # re-index would silently apply the old ownership bounds to other Gaussians.  merge_moments() before the surgery clears the bounds.
            raise RuntimeError("ShardedAdam: the optimizer state was re-indexed (select_state) between two steps without merge_moments(); wire "
                               "strategy.before_reindex = sharded.merge_moments (trainer.Trainer does) or call it before the surgery")
        self._gen = gen
        if self._bounds is not None and self._bounds[0] != n:
            # the partition moves: bring every rank's moments up to date under the OLD partition before rows change owner.
            # (rows appended or reset by the densification strategy since are zero on every rank; merging zeros keeps them zero)
            n_old, lo_old, hi_old = self._bounds
            # only GROWTH by appended rows is supported here: after a shrink / re-index (MCMC.remove_gaussians, select_state) the old
            # bounds no longer name the same Gaussians — checked BEFORE any collective touches the moments (call merge_moments() ahead
            # of such a surgery instead)
            for st in self.opt.state.values():
                if isinstance(st, dict):
                    for k in ("exp_avg", "exp_avg_sq"):
                        if st[k].shape[0] < n_old:
                            raise RuntimeError("ShardedAdam: the model shrank or was re-indexed (%d -> %d rows) between two steps; call "
                                               "merge_moments() before the surgery so every rank holds complete moments" % (n_old, st[k].shape[0]))
            for st in self.opt.state.values():
                if isinstance(st, dict):
                    for k in ("exp_avg", "exp_avg_sq"):
                        grown = st[k].shape[0] - n_old   # rows appended after the last step: owned by nobody yet, identical everywhere
                        head = st[k][:n_old]
                        self._merge_owned_rows(head, lo_old, hi_old)
                        assert grown >= 0
        self._bounds = (n, lo, hi)
        for p in params:
            self._reduce_mean(p.grad, lo, hi, even)
        self.bucket.last_reduced_bytes = self.bucket.nbytes()
        self.opt.step(iteration, rows=(lo, hi))
        for p in params:
            self._gather_rows(p.data, lo, hi, even)


class ColorGradExchange:
    """Gradient exchange of the camera-sharded step that never moves the SH gradient.

    81 % of the gradient bucket is v_sh [N,K,3] (48 of 59 floats per Gaussian at degree 3), and every rank's v_sh is an outer product:
    v_sh[g] = basis(direction of camera r to Gaussian g) (x) v_colors_r[g] — K*3 floats built from 3.  So the ranks exchange the THREE
    colour-gradient floats per (camera, Gaussian) with one all-gather (12 B per Gaussian per rank, every rank sends its slice to its 7
    peers over its 7 xGMI links at once: the pattern the fully connected node is built for), and every rank runs the fused SH backward
    over ALL cameras of the step (gsx_sh_colors_bwd, C = world, pre-masked colour gradients): the 192 B SH row of a Gaussian is
    written once either way, the extra cameras cost 12 B of reads and a basis evaluation each.  The remaining 11 floats per Gaussian
    (means from the blend, scaling, rotation, opacity) are all-reduced as before, while the SH backward runs.
        per rank, 1 M Gaussians, 8 GPUs:  dense all-reduce 2 * 7/8 * 236 MB = 413 MB sent  ->  84 MB (all-gather) + 77 MB (all-reduce of 44 MB)
    Every rank computes v_sh from the same gathered bits in the same camera order, so the replicas stay bit-identical (the dense
    all-reduce only promises that through the collective).  The mean over the cameras is applied to the colour gradients before
    they travel (1 / world), the all-reduce averages as GradBucket does.

    Wiring: `sinks = bucket.sinks(); sinks["_color_exchange"] = xch; xch.begin_step(viewmats_all)` before the render's backward;
    rasterizer.GutRenderFunction.backward then calls xch.sh_backward(...) in place of the local SH backward, and the caller calls
    xch.finish() before the optimizer reads the gradients."""

    def __init__(self, bucket, names=("means", "sh", "scaling_raw", "rotation_raw", "opacity_raw"), sh_bwd_fn=None):
        self.bucket, self.names = bucket, list(names)
        self.sh_bwd_fn = sh_bwd_fn   # tests: a CPU stand-in for ops.sh_colors_bwd (same signature)
        self._viewmats = None
        self._pending = []
        self._buf = self._tmp = None
        # the spans of the flat bucket that are NOT the SH gradient: one all-reduce per contiguous run (one when sh comes first or last)
        i_sh = self.names.index("sh")
        runs, cur = [], None
        for i, (p, o) in enumerate(zip(bucket.params, bucket.offsets)):
            if i == i_sh:
                cur = None
                continue
            end = bucket.offsets[i + 1] if i + 1 < len(bucket.offsets) else bucket.flat.numel()
            if cur is None:
                cur = [o, end]
                runs.append(cur)
            else:
                cur[1] = end
        self._runs = [(a, b) for a, b in runs]

    def begin_step(self, viewmats_all):
        """viewmats_all: [world,4,4] world->camera matrices of the step's cameras, row r = the camera rank r renders (every rank knows
        the step's batch: the camera schedule is a function of the iteration)."""
        assert viewmats_all.shape[0] == dist.get_world_size()
        self._viewmats = viewmats_all.contiguous()

    def sh_backward(self, sh_degree, means, sh, colors, v_colors, v_means_blend, sink_sh, sink_means, sh_adam=None):
        """colors / v_colors: [1,N,3] of this rank's camera (post-clamp colours, gradient from the blend); v_means_blend: the blend's
        gradient w.r.t. the means.  Returns (v_sh, v_means) = the sinks, both complete (mean over the step's cameras) when finish()
        has returned."""
        world, rank = dist.get_world_size(), dist.get_rank()
        n = means.shape[0]
        if self._buf is None or self._buf.shape[1] != n or self._buf.device != means.device:
            self._buf = torch.empty(world, n, 3, dtype=means.dtype, device=means.device)
            self._tmp = torch.empty(n, 3, dtype=means.dtype, device=means.device)
        mine = self._buf[rank]
        # clamp_min(x + 0.5, 0) passes the gradient where the colour is positive; a Gaussian the camera does not see has a zero colour
        # row (sh_colors_fwd) and a zero gradient row (the blend's gather kernel): its row stays zero.  1 / world = mean over the cameras.
        torch.mul(v_colors.reshape(n, 3), 1.0 / world, out=mine)
        mine.mul_(colors.reshape(n, 3) > 0)
        gather = dist.all_gather_into_tensor(self._buf.view(-1), mine.view(-1), async_op=True)   # first: the SH backward waits for it
        sink_means.copy_(v_means_blend.reshape(sink_means.shape))
        self._pending = [_MeanHandle(self.bucket.flat[a:b]) for a, b in self._runs]   # means | scaling | rotation | opacity, under the SH backward
        gather.wait()
        fn = self.sh_bwd_fn
        if sh_adam is not None and fn is None:
            # the SH gradient is complete on every rank inside this kernel, so the Adam step of the SH tensor is applied right there
            # (optim.FusedAdam.begin_fused_sh_step): every rank performs the identical update on identical bits
            from . import ops
            ops.sh_colors_bwd_adam(sh_degree, means, self._viewmats, sh, None, None, self._buf, None, self._tmp, *sh_adam)
        else:
            if fn is None:
                from . import ops
                fn = ops.sh_colors_bwd
            fn(sh_degree, means, self._viewmats, sh, None, None, self._buf, None, sink_sh, self._tmp)
        self.bucket.last_reduced_bytes = self._buf.numel() * 4 + sum(b - a for a, b in self._runs) * 4
        return sink_sh, sink_means

    def finish(self):
        """All gradients are complete on the current stream after this: the blend part of the means gradient (averaged over the ranks)
        plus the direction part the SH backward produced for all cameras."""
        for h in self._pending:
            h.wait()
        if self._pending:
            i = self.names.index("means")
            self.bucket.params[i].grad.add_(self._tmp)
        self._pending = []


class CameraBatchAccumulator:
    """C cameras per optimizer step on ONE GPU: the single-process counterpart of the camera-sharded step (BASELINE configs[3] on one
    MI355X — the denominator of `north_star`'s scaling target).  The reference renders one camera per iteration (trainer.cpp:917-922);
    a batch of C cameras is that loop body C times with the gradients averaged and ONE optimizer step, which is also exactly what N
    ranks x 1 camera compute.  The same arithmetic as ColorGradExchange without the collectives:

      camera c = 0 .. C-1 (each a full render + loss + backward through rasterize_fused with this object as sinks["_color_exchange"]):
          colour gradient of the camera, masked by the colour clamp, times 1/C  ->  row c of a [C,N,3] buffer         (12 B / Gaussian)
          the blend's means gradient and the scaling / rotation / opacity sinks  ->  added into an 11-float accumulator  (44 B / Gaussian)
      after camera C-1: ONE SH backward over all C cameras (fused with the SH tensor's Adam step when the caller passes one), the
          accumulator / C goes back into the bucket: the 192 B SH gradient row of a Gaussian is produced once per step, not once per camera.

    Wiring: acc = CameraBatchAccumulator(bucket, names, C); sinks["_color_exchange"] = acc; acc.begin_step(viewmats [C,4,4]);
    C x (rasterize_fused(..., grad_sinks=sinks) -> loss -> backward); acc.finish(); optimizer step.  sinks["_sh_adam"] may be set for
    every backward: it is used by the last camera's only."""

    def __init__(self, bucket, names=("means", "sh", "scaling_raw", "rotation_raw", "opacity_raw"), cameras=8, sh_bwd_fn=None):
        self.bucket, self.names, self.C = bucket, list(names), int(cameras)
        self.sh_bwd_fn = sh_bwd_fn   # tests: a CPU stand-in for ops.sh_colors_bwd (same signature)
        self._viewmats = None
        self._buf = self._tmp = self._acc = None
        self._c = 0
        self._done = False
        # the spans of the flat bucket that are not the SH gradient, as in ColorGradExchange (everything the per-camera backward overwrites)
        i_sh = self.names.index("sh")
        self._spans = []
        for i, o in enumerate(bucket.offsets):
            if i == i_sh:
                continue
            end = bucket.offsets[i + 1] if i + 1 < len(bucket.offsets) else bucket.flat.numel()
            if self._spans and self._spans[-1][1] == o:
                self._spans[-1][1] = end
            else:
                self._spans.append([o, end])

    def begin_step(self, viewmats_all):
        assert viewmats_all.shape[0] == self.C
        self._viewmats = viewmats_all.contiguous()
        self._c = 0
        self._done = False

    def sh_backward(self, sh_degree, means, sh, colors, v_colors, v_means_blend, sink_sh, sink_means, sh_adam=None):
        """Same contract as ColorGradExchange.sh_backward; called once per camera of the batch, in the order of begin_step's viewmats."""
        n, c = means.shape[0], self._c
        assert self._viewmats is not None and c < self.C, "CameraBatchAccumulator: begin_step() first, C backwards per step"
        flat = self.bucket.flat
        if self._buf is None or self._buf.shape[1] != n or self._buf.device != means.device:
            self._buf = torch.empty(self.C, n, 3, dtype=means.dtype, device=means.device)
            self._tmp = torch.empty(n, 3, dtype=means.dtype, device=means.device)
            self._acc = [torch.empty(b - a, dtype=flat.dtype, device=flat.device) for a, b in self._spans]
        torch.mul(v_colors.reshape(n, 3), 1.0 / self.C, out=self._buf[c])
        self._buf[c].mul_(colors.reshape(n, 3) > 0)
        sink_means.copy_(v_means_blend.reshape(sink_means.shape))
        for acc, (a, b) in zip(self._acc, self._spans):   # means | scaling | rotation | opacity of THIS camera
            if c == 0:
                acc.copy_(flat[a:b])
            else:
                acc.add_(flat[a:b])
        self._c = c + 1
        if self._c < self.C:
            return sink_sh, sink_means
        fn = self.sh_bwd_fn
        if sh_adam is not None and fn is None:
            from . import ops
            ops.sh_colors_bwd_adam(sh_degree, means, self._viewmats, sh, None, None, self._buf, None, self._tmp, *sh_adam)
        else:
            if fn is None:
                from . import ops
                fn = ops.sh_colors_bwd
            fn(sh_degree, means, self._viewmats, sh, None, None, self._buf, None, sink_sh, self._tmp)
        for acc, (a, b) in zip(self._acc, self._spans):
            torch.mul(acc, 1.0 / self.C, out=flat[a:b])
        self.bucket.params[self.names.index("means")].grad.add_(self._tmp)   # the direction part, summed over the cameras (colour gradients carry 1/C)
        self._done = True
        return sink_sh, sink_means

    def finish(self):
        assert self._done, "CameraBatchAccumulator.finish(): %d of %d cameras of the step have run their backward" % (self._c, self.C)
        self._viewmats = None


class ListsAgreement:
    """Guarded intersection lists (rasterizer.rasterize_fused(guarded=True)) under N ranks: a frame whose lists overflowed on ANY rank
    is rendered again on EVERY rank, so the collectives of the step stay matched and the replicas identical.  The verdicts meet on
    the HOST while every GPU still has the forward, the loss and the blend backward queued: no device-side collective, no stream
    synchronisation.

    Transport.  Ranks of one node (the camera-sharded step is a single-node design: `north_star`) meet in a SHARED-MEMORY page: every rank
    publishes (call number, verdict) in its own cache line and reads the others' — a few microseconds however many ranks there are.  (Round 4
    used a MIN all-reduce of one int over a gloo group; measured with tools/lists_agree_time.py on an 8-core box: 0.25 ms with 2 ranks,
    1.1 ms with 8 — as long as the iteration it guards.)  Ranks that do not share a host (or GSX_AGREE=gloo) keep the gloo all-reduce.
    A rank that fails before it reaches its vote must still vote (`agree(False)` from the caller's exception path) or its peers wait: the
    wait gives up after `timeout_s` with an error naming the silent ranks instead of hanging.
    Wiring: `sinks["_lists_agree"] = ListsAgreement()` (constructed collectively on every rank)."""

    def __init__(self, timeout_s=120.0):
        self.group = None
        self.timeout_s = float(timeout_s)
        self._shm = self._slots = None
        self._calls = 0
        self.transport = "none"
        self.disagreements = 0   # iterations repeated because some OTHER rank overflowed
        self._flag = torch.zeros(1, dtype=torch.int32)
        if not active():
            return
        import datetime
        import socket
        with _stdout_to_stderr():
            self.group = dist.group.WORLD if dist.get_backend() == "gloo" else dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=self.timeout_s))
        self.transport = "gloo"
        world, rank = dist.get_world_size(), dist.get_rank()
        if os.environ.get("GSX_AGREE", "shm") != "shm":
            return
        # Every collective below is executed by EVERY rank whatever happens locally (a rank that cannot create or map the page reports it and all
        # ranks fall back to gloo together): an exception on one rank must not leave its peers waiting in a collective it skipped.
        from multiprocessing import shared_memory
        import numpy as np
        hosts = [None] * world
        dist.all_gather_object(hosts, socket.gethostname(), group=self.group)
        same_host = all(h == hosts[0] for h in hosts)
        name = [None]
        if rank == 0 and same_host:
            try:
                self._shm = shared_memory.SharedMemory(create=True, size=64 * world)   # one cache line per rank
                self._shm_owner = True   # (remembered here: close() may run after the process group is gone, e.g. from __del__ at interpreter exit)
                self._shm.buf[:64 * world] = bytes(64 * world)
                name[0] = self._shm.name
            except Exception:  # noqa: BLE001  (no /dev/shm, a sandbox without shared memory ...)
                self._shm, name[0] = None, None
        dist.broadcast_object_list(name, src=0, group=self.group)
        mapped = 0
        if name[0] is not None:
            try:
                if rank != 0:
                    self._shm = shared_memory.SharedMemory(name=name[0])
                    try:   # the segment belongs to rank 0: this process's resource tracker must not unlink it at exit
                        from multiprocessing import resource_tracker
                        resource_tracker.unregister(self._shm._name, "shared_memory")
                    except Exception:  # noqa: BLE001
                        pass
                self._slots = np.ndarray((world, 8), dtype=np.int64, buffer=self._shm.buf)   # [rank][0 / 1] = (call number << 1) | verdict of its latest even / odd call (see __call__)
                mapped = 1
            except Exception:  # noqa: BLE001
                self._slots = None
        flag = torch.tensor([mapped], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)   # every rank has mapped the page — or nobody uses it (also the barrier before rank 0 can go away)
        if int(flag[0]) == 1:
            self.transport = "shm"
        else:
            self._slots = None
            self.close()

    def __call__(self, ok):
        if self.group is None:
            return bool(ok)
        if self._slots is not None:
            import time
            self._calls += 1
            n, me = self._calls, dist.get_rank()
            # Slot of a rank (one 64 B cache line): words 0 / 1 = its latest even / odd call as ONE aligned int64, (call number << 1) | verdict.
            # Number and verdict travel in a single 8-byte store, so a peer that sees the number has the verdict with it: nothing relies on the
            # order of two stores (x86 keeps it, aarch64 does not: ADVICE r05).  A rank leaves call n only after it has seen EVERY rank's word
            # reach n, so no rank can be more than one call ahead of the slowest one: while a slow rank still reads the words of call n, a fast
            # one may already have published call n + 1 — into the OTHER word; word n & 1 is rewritten at call n + 2, which nobody reaches
            # before everybody has left call n.
            w = n & 1
            self._slots[me, w] = (n << 1) | (1 if ok else 0)
            agreed, deadline, spins = True, None, 0
            for r in range(dist.get_world_size()):
                while True:
                    word = int(self._slots[r, w])
                    if (word >> 1) >= n:
                        break
                    spins += 1
                    if spins & 0xFF == 0:
                        now = time.monotonic()
                        deadline = deadline or now + self.timeout_s
                        if now > deadline:
                            raise RuntimeError("ListsAgreement: rank %d did not vote on call %d within %.0f s (it failed before its vote, or hangs)" % (r, n, self.timeout_s))
                    if spins > 20000:
                        time.sleep(0.0002)   # a peer that is milliseconds late is not coming back soon: stop pinning a core
                    elif spins > 256:
                        time.sleep(0)   # a peer that is ~100 us late may be waiting for a core (oversubscribed host): yield instead of spinning on
                agreed = agreed and bool(word & 1)
        else:
            self._flag[0] = 1 if ok else 0
            dist.all_reduce(self._flag, op=dist.ReduceOp.MIN, group=self.group)
            agreed = bool(int(self._flag[0]))
        if ok and not agreed:
            self.disagreements += 1
        return agreed

    def close(self):
        if self._shm is not None:
            self._slots = None
            try:
                self._shm.close()
                if getattr(self, "_shm_owner", False):
                    self._shm.unlink()
            except Exception:  # noqa: BLE001
                pass
            self._shm = None

    def __del__(self):
        self.close()


def shard_cameras(cameras, rank, world):
    """Camera i of the step's batch goes to rank i % world (one camera per GPU when len == world)."""
    return [c for i, c in enumerate(cameras) if i % world == rank]
