#!/usr/bin/env python
"""bench.py — the hot path's headline benchmark (BASELINE.json): forward + backward of the `--gut`
rasterizer at 1 M Gaussians / SH degree 3 / 1920x1080 (configs[1], "S-1M"), on N GPUs of one node.

A step = one pass of the hot path over one camera per rank:
    activations -> projection_ut -> SH colours (+0.5, clamp) -> intersect_tile (+sort) -> intersect_offset -> blend fwd
    -> L1 loss -> blend bwd -> SH bwd -> activation Jacobians [-> gradient all-reduce when N > 1]
Default: the fused glue kernels of rasterize_fused (gradients written straight into the flat all-reduce bucket);
--unfused runs the reference-style chain of torch ops around the seven gsplat operators.
Inputs are resident in HBM before the timed region.  N > 1: launched by torch.distributed.run, one rank per
GPU over RCCL; every rank renders its own camera of the step's batch (cameras on a small orbit around the
cfg2 pose so per-GPU work stays fixed: weak scaling) and the per-Gaussian gradients (59 fp32 / Gaussian, one
flat bucket) are all-reduced after backward.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel, live HIP-event
timing) and `cpu_baseline` (the CPU oracle timed on one full frame of the same workload, ~10 s on 8 cores).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s measured copy


def algorithmic_bytes(N, C, I, P, tiles, deg, K):  # noqa: E741  
    nb = (deg + 1) ** 2
    return {
        "projection_ut_3dgs_fused": 76 * N * C,
        "spherical_harmonics_fwd": N * C * (12 + 12 * nb + 1 + 12),
        "spherical_harmonics_bwd": N * C * (12 + 12 * nb + 1 + 12 + 12 * K + 12),
        # fused variants: means instead of dirs, radii (8 B) instead of the mask; bwd also reads colours and v_means
        "sh_colors_fwd": N * C * (12 + 12 * nb + 8 + 12),
        "sh_colors_bwd": N * C * (12 + 12 * nb + 8 + 12 + 12 + 12 * K + 24),
        "splat_activations_fwd": N * (40 + 44),
        "splat_activations_bwd": N * (40 + 44 + 40),
        "intersect_tile": 20 * N * C + 12 * N * C + (28 * N * C + 12 * I) + 144 * I,
        "intersect_offset": 8 * I + 4 * tiles,
        # binned variant = both reference ops in one pipeline: priced at the reference's algorithmic bytes for the two
        "intersect_tile_binned": 20 * N * C + 12 * N * C + (28 * N * C + 12 * I) + 144 * I + 8 * I + 4 * tiles,
        "rasterize_to_pixels_from_world_3dgs_fwd": 60 * I + 20 * P + 4 * tiles,
        "rasterize_to_pixels_from_world_3dgs_bwd": 172 * I + 24 * P,
        # fused loss: fwd reads render + gt (24 B/px), writes 3 chained derivative maps x 3 channels (36 B/px);
        # bwd reads those + render + gt, writes v_render (12 B/px)
        "photometric_loss_fwd": 60 * P * C,
        "photometric_loss_bwd": 72 * P * C,
    }


class OpTimer:
    """Brackets every gsplat op with HIP events on the stream the kernels are launched on (torch's current
    stream: the shim launches on c10's current HIP stream)."""

    def __init__(self, ops_mod):
        self.ops = ops_mod
        self.names = ["projection_ut_3dgs_fused", "spherical_harmonics_fwd", "spherical_harmonics_bwd", "intersect_tile",
                      "intersect_offset", "rasterize_to_pixels_from_world_3dgs_fwd", "rasterize_to_pixels_from_world_3dgs_bwd",
                      "sh_colors_fwd", "sh_colors_bwd", "splat_activations_fwd", "splat_activations_bwd",
                      "photometric_loss_fwd", "photometric_loss_bwd", "intersect_tile_binned"]
        self.orig = {n: getattr(ops_mod, n) for n in self.names}
        self.events = {n: [] for n in self.names}
        self.enabled = False
        self.only = None  # restrict the bracketing to these ops (every event record costs a ~5 us bubble on the stream)
        for n in self.names:
            setattr(ops_mod, n, self._wrap(n))

    def _wrap(self, name):
        fn = self.orig[name]

        def wrapped(*a, **k):
            if not self.enabled or (self.only is not None and name not in self.only):
                return fn(*a, **k)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = fn(*a, **k)
            e.record()
            self.events[name].append((s, e))
            return r
        return wrapped

    def reset(self):
        self.events = {n: [] for n in self.names}

    def mean_ms(self):
        out = {}
        for n, evs in self.events.items():
            if evs:
                out[n] = sum(s.elapsed_time(e) for s, e in evs) / len(evs)
        return out


def cpu_baseline(scene, threads):
    """The CPU oracle (a restatement of the reference kernels: kind "port", OpenMP over tiles / Gaussians) timed
    on ONE full step of the same workload (the whole S-1M frame, forward + backward)."""
    import numpy as np
    from oracle import oracle
    from tests.helpers import oracle_pipeline
    os.environ.setdefault("OMP_NUM_THREADS", str(threads))
    H, W, deg = scene["height"], scene["width"], scene["sh_degree"]
    rng = np.random.default_rng(0)
    v_rc = rng.standard_normal((1, H, W, 3)).astype(np.float32)
    v_ra = np.zeros((1, H, W, 1), np.float32)
    t0 = time.perf_counter()
    o = oracle_pipeline(scene, v_render_colors=v_rc, v_render_alphas=v_ra)
    oracle.sh_bwd(deg, o["dirs"].reshape(-1, 3), scene["sh"].numpy(), o["masks"].reshape(-1), o["v_colors"].reshape(-1, 3), True)
    dt = time.perf_counter() - t0
    return dt, int(o["flatten_ids"].shape[0])


def cpu_reference_stages(scene_small):
    """The stages the reference's own CPU code (tests/torch_impl.cpp compiled unmodified into oracle/_ref) implements —
    EWA projection (not the UT projection of the hot path), SH evaluation, tile intersection (a serial loop upstream) — timed on
    the reference's CPU-runnable case (cfg1 / S-small, 10 k Gaussians @256x256).  There is no CPU rasterizer upstream to time."""
    import numpy as np
    from oracle import ref
    if not ref.available():
        return None
    f = lambda k: np.ascontiguousarray(scene_small[k].numpy(), np.float32)  # noqa: E731
    means, quats, scales = f("means"), f("quats"), f("scales")
    vm, K = f("viewmat")[None], f("K")[None]
    W, H = scene_small["width"], scene_small["height"]
    out = {}
    t0 = time.perf_counter()
    radii, m2d, dep, _ = ref.ewa_projection(means, quats, scales, vm, K, W, H)
    out["ewa_projection_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
    dirs = means - np.linalg.inv(vm[0].astype(np.float64))[:3, 3].astype(np.float32)
    t0 = time.perf_counter()
    ref.spherical_harmonics(scene_small["sh_degree"], dirs, f("sh"))
    out["spherical_harmonics_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
    t0 = time.perf_counter()
    ref.isect_tiles(m2d, radii, dep, 16, (W + 15) // 16, (H + 15) // 16, True)
    out["isect_tiles_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
    out["workload"] = "S-small (BASELINE configs[0]): %d Gaussians, %dx%d" % (means.shape[0], W, H)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scene", default="1m", choices=["small", "1m", "5m"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default: nccl = RCCL)")
    ap.add_argument("--unfused", action="store_true", help="reference-style glue (one torch op per activation / SH pre-post step)")
    ap.add_argument("--sparse-allreduce", action="store_true", help="exchange only the gradient rows of Gaussians some camera saw")
    ap.add_argument("--l1-loss", action="store_true", help="plain torch L1 loss instead of the reference's fused L1 + SSIM loss")
    ap.add_argument("--no-train-iter", action="store_true", help="skip the extra full-iteration timing (loss + backward + Adam)")
    args = ap.parse_args()

    import gsx  # noqa: F401
    from gsx import distributed as gdist
    from gsx import loss as gloss
    from gsx import ops, optim, rasterizer, scenes

    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    if os.environ.get("GSX_BENCH_ALL_RANKS_ON_DEVICE0"):  # functional test of the N>1 code path on a 1-GPU box (gloo)
        os.environ["LOCAL_RANK"] = "0"
    rank, local_rank, world = gdist.init_from_env(args.backend)
    assert world == max(1, args.gpus), "--gpus must equal WORLD_SIZE (launch N>1 with torch.distributed.run)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    scene = {"small": scenes.scene_small, "1m": scenes.scene_1m, "5m": scenes.scene_5m}[args.scene]()
    N, W, H, deg = scene["means"].shape[0], scene["width"], scene["height"], scene["sh_degree"]
    model = scenes.to_splat_data(scene, dev)
    for p in model.params():
        p.requires_grad_(True)
    bucket = gdist.GradBucket(model.params())
    # camera of this rank: cfg2 pose on a 5 cm orbit (rank 0 of a 1-GPU run = exactly the cfg2 camera)
    vm = scene["viewmat"].clone()
    if world > 1:
        a = 2 * math.pi * rank / world
        vm[0, 3], vm[1, 3] = 0.05 * math.cos(a), 0.05 * math.sin(a)
    cam = rasterizer.Camera(viewmat=vm.to(dev), K=scene["K"].to(dev), width=W, height=H)
    bg = scene["background"].to(dev)
    g = torch.Generator().manual_seed(1234 + rank)
    target = torch.rand(3, H, W, generator=g).to(dev)
    fused_loss = not (args.l1_loss or args.unfused)

    timer = OpTimer(ops)
    state = {}

    sinks = bucket.sinks()

    def step():
        # fused glue: gradients are written straight into the flat bucket (no zero fill, no AccumulateGrad adds)
        out = rasterizer.rasterize_fused(cam, model, bg, grad_sinks=sinks) if not args.unfused else None
        if args.unfused:
            bucket.zero_()
            out = rasterizer.rasterize(cam, model, bg)
        # the reference's photometric loss (trainer.cpp:103-127): 0.8 L1 + 0.2 (1 - SSIM), fused on the blend's own layout
        loss = gloss.photometric_loss(out.render_hwc, target, 0.2) if fused_loss else (out.image - target).abs().mean()
        loss.backward()
        if world > 1:
            if args.sparse_allreduce and not args.unfused:
                # rows no camera of the step sees are zero on every rank: only the union of the visible rows travels (in S-1M every
                # camera sees ~all Gaussians, so this falls back to the dense collective after one probe)
                bucket.all_reduce_mean_rows((out.aux["radii_full"] > 0).all(-1))
            else:
                bucket.all_reduce_mean()
        state["n_isects"] = out.n_isects

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # Timed region: only the two blend ops (the roofline kernels) are bracketed with HIP events — each event record opens a
    # ~5 us bubble on the stream, 22 ops x 2 events would cost ~0.1 ms per step.  The per-op table of the other ops is
    # measured in a separate pass after the timed region.
    timer.enabled = True
    timer.only = {"rasterize_to_pixels_from_world_3dgs_fwd", "rasterize_to_pixels_from_world_3dgs_bwd"}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    n_isects_timed = state["n_isects"]  # (the train-iteration leg below moves the Gaussians: later steps have another count)
    blend_ms = timer.mean_ms()
    timer.reset()
    timer.only, timer.enabled = None, True   # per-op pass (outside the timed region)
    for _ in range(min(args.steps, 10)):
        step()
    torch.cuda.synchronize()
    timer.enabled = False
    all_ms = timer.mean_ms()
    all_ms.update(blend_ms)                  # the blend ops keep their timed-region figures
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # second part of the BASELINE metric ("train iters/s"): the same step followed by the fused Adam update of all six
    # parameter groups (optim.FusedAdam, reference lrs) — timed separately, after and outside the K contract steps.
    train_iter_ms = None
    if not args.no_train_iter and not args.unfused:
        # (.grad of every parameter already is its view of the flat bucket the backward writes into)
        opt = optim.FusedAdam.for_splat_data(model)
        it = [1000]

        def train_iter():
            it[0] += 1
            step()
            opt.step(it[0])

        for _ in range(2):
            train_iter()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            train_iter()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        tt = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        train_iter_ms = float(tt.item()) / args.steps * 1e3

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        I = n_isects_timed  # noqa: E741
        P = W * H
        tiles = ((W + 15) // 16) * ((H + 15) // 16)
        K = (deg + 1) ** 2
        ab = algorithmic_bytes(N, 1, I, P, tiles, deg, K)
        op_ms = all_ms
        kernels = {}
        for n, ms in op_ms.items():
            gbs = ab[n] / (ms * 1e-3) / 1e9
            kernels[n] = {"ms": round(ms, 4), "algorithmic_bytes": int(ab[n]), "GBps": round(gbs, 1),
                          "frac_hbm": round(gbs / HBM_PEAK_GBS, 4)}
        blend = [n for n in op_ms if n.startswith("rasterize_to_pixels")]
        dom = max(blend or list(op_ms), key=lambda n: op_ms[n])
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(dom)
            except Exception:
                traffic = None
        roofline = {"kernel": dom, "bound": "hbm", "achieved": kernels[dom]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": kernels[dom]["frac_hbm"], "traffic": traffic,
                    "algorithmic_bytes_per_launch": kernels[dom]["algorithmic_bytes"], "avg_launch_ms": kernels[dom]["ms"]}
        result = {
            "metric": "train iters/s: fwd+bwd frames/s through the gut rasterizer, 1M Gaussians @1080p SH3",
            "value": round(world * args.steps / elapsed, 4),
            "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "S-1M (BASELINE configs[1]): 1M random Gaussians, SH deg 3, 1920x1080, fwd+bwd, "
                                   "one camera per GPU" if args.scene == "1m" else args.scene,
                       "n_gaussians": N, "width": W, "height": H, "sh_degree": deg, "n_isects": I,
                       "cameras_per_step": world, "grad_allreduce_bytes": int(getattr(bucket, "last_reduced_bytes", 0)) if world > 1 else 0,
                       "grad_bucket_bytes": bucket.nbytes()},
            "gaussians_x_pixels_per_s": round(world * N * P / (elapsed / args.steps), 1),
            "pairs_per_s_fwd": round(256.0 * I / (op_ms.get("rasterize_to_pixels_from_world_3dgs_fwd", float("nan")) * 1e-3), 1),
            "roofline": roofline,
            "kernels": kernels,
            "loss": "fused L1 + SSIM (lambda 0.2)" if fused_loss else "torch L1",
        }
        if train_iter_ms is not None:
            result["train_iter"] = {"ms": round(train_iter_ms, 4), "iters_per_s": round(1e3 / train_iter_ms, 3),
                                    "frames_per_s": round(world * 1e3 / train_iter_ms, 3),
                                    "what": "render + fused L1/SSIM loss + backward + grad all-reduce + fused Adam (6 groups)"}
        if world == 1 and not args.no_cpu_baseline:
            threads = os.cpu_count() or 1
            dt, i_cpu = cpu_baseline(scene, threads)
            try:
                stages = cpu_reference_stages(scenes.scene_small())
            except Exception as e:  # the reference build is optional on the GPU box
                stages = {"error": str(e)[:200]}
            if stages:
                result["cpu_reference_stages"] = stages
            result["cpu_baseline"] = {"value": round(1.0 / dt, 5), "unit": "frames/s", "cores": threads, "kind": "port",
                                      "sample": "oracle (CPU restatement, OpenMP) forward+backward of ONE full frame of the "
                                                "same workload (%d isects): %.2f s" % (i_cpu, dt)}
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
