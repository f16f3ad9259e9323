"""N>1 path on CPU: two processes over gloo exercise the camera sharding + the flat gradient bucket all-reduce
(gaussian-splatting-cuda_amd/distributed.py).  Per-rank gradients come from the CPU oracle (tests may use it),
one camera per rank; after the collective every rank must hold the mean of both cameras' gradients."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _camera_grads(rank):
    """Oracle gradients of sum(image * w) for camera `rank` of a 2-camera orbit around a small scene."""
    from gsx import scenes
    from tests.helpers import oracle_pipeline
    sc = scenes.scene_small(seed=5, N=300)
    sc["width"] = sc["height"] = 48
    sc["K"] = scenes.intrinsics(40.0, 40.0, 24.0, 24.0)
    vm = torch.eye(4)
    vm[0, 3] = 0.05 if rank == 0 else -0.05
    sc["viewmat"] = vm
    rng = np.random.default_rng(7)
    v_rc = rng.standard_normal((1, 48, 48, 3)).astype(np.float32)
    v_ra = rng.standard_normal((1, 48, 48, 1)).astype(np.float32)
    o = oracle_pipeline(sc, v_render_colors=v_rc, v_render_alphas=v_ra)
    return [o["v_means"], o["v_quats"], o["v_scales"], o["v_opacities"].reshape(-1, 1)]


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import gsx  # noqa: F401
    from gsx import distributed as gdist
    r, lr, w = gdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    grads = _camera_grads(rank)
    params = [torch.zeros(g.shape, dtype=torch.float32, requires_grad=True) for g in grads]
    bucket = gdist.GradBucket(params)
    assert bucket.flat.numel() >= sum(g.size for g in grads)   # views start on 256 B boundaries
    for p, g in zip(params, grads):
        assert p.grad.data_ptr() >= bucket.flat.data_ptr()          # .grad is a view into the bucket
        p.grad.add_(torch.from_numpy(g))
    bucket.all_reduce_mean()
    cams = list(range(5))
    assert gdist.shard_cameras(cams, rank, world) == [c for c in cams if c % world == rank]
    np.save(os.path.join(out_dir, "bucket_%d.npy" % rank), np.concatenate([p.grad.numpy().reshape(-1) for p in params]))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_bucket_all_reduce_two_ranks(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    b0 = np.load(tmp_path / "bucket_0.npy")
    b1 = np.load(tmp_path / "bucket_1.npy")
    assert np.array_equal(b0, b1)                                    # replicas stay identical
    expect = 0.5 * (np.concatenate([g.reshape(-1) for g in _camera_grads(0)]) +
                    np.concatenate([g.reshape(-1) for g in _camera_grads(1)]))
    np.testing.assert_allclose(b0, expect, rtol=1e-6, atol=1e-7)


def test_single_process_is_a_noop():
    import gsx  # noqa: F401
    from gsx import distributed as gdist
    p = torch.ones(4, 3, requires_grad=True)
    b = gdist.GradBucket([p])
    p.grad.fill_(2.0)
    assert b.all_reduce_mean() is None and torch.all(p.grad == 2.0)
    b.zero_()
    assert torch.all(p.grad == 0)


def _rows_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import gsx  # noqa: F401
    from gsx import distributed as gdist
    gdist.init_from_env(backend="gloo")
    N = 1000
    g = torch.Generator().manual_seed(100 + rank)
    shapes = [(N, 3), (N, 16, 3), (N, 3), (N, 4), (N, 1)]
    visible = torch.rand(N, generator=g) < 0.3                      # this rank's camera sees ~30 % of the Gaussians
    grads = [torch.randn(s, generator=g) * visible.view(-1, *([1] * (len(s) - 1))) for s in shapes]
    for mode in ("rows", "dense", "rows_fallback", "tail_head"):
        params = [torch.zeros(s, requires_grad=True) for s in shapes]
        bucket = gdist.GradBucket(params)
        for p, gr in zip(params, grads):
            p.grad.copy_(gr)
        if mode == "dense":
            bucket.all_reduce_mean()
        elif mode == "tail_head":   # the overlapped exchange: last three parameters first (asynchronously), then the head
            h = bucket.all_reduce_mean_tail_async(2)
            bucket.all_reduce_mean_head(h)
        else:
            bucket.all_reduce_mean_rows(visible, dense_above=0.75 if mode == "rows" else 0.1)
        if mode == "rows":
            assert bucket.last_reduced_bytes < 0.7 * bucket.nbytes()
        np.save(os.path.join(out_dir, "%s_%d.npy" % (mode, rank)), np.concatenate([p.grad.numpy().reshape(-1) for p in params]))
    dist.barrier()
    dist.destroy_process_group()


def test_visible_row_all_reduce_equals_dense(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_rows_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    dense = np.load(tmp_path / "dense_0.npy")
    for mode in ("rows", "rows_fallback", "tail_head"):
        for r in range(world):
            got = np.load(tmp_path / ("%s_%d.npy" % (mode, r)))
            assert np.array_equal(got, dense), (mode, r)             # bit-identical to the dense collective on 2 ranks
    assert np.array_equal(np.load(tmp_path / "dense_1.npy"), dense)


class _RowSGD:
    """Stand-in for optim.FusedAdam in the CPU test of ShardedAdam (the fused Adam kernel needs a GPU): same step(iteration, rows=)
    contract, elementwise update, so a row-sharded step followed by the all-gather must equal the replicated step bit for bit."""

    def __init__(self, params, lr=0.1):
        self.params, self.lr, self.calls = params, lr, []

    def step(self, iteration, rows=None):
        self.calls.append(rows)
        for p in self.params:
            sl = slice(None) if rows is None else slice(rows[0], rows[1])
            p.data[sl] -= self.lr * p.grad[sl] * (1.0 + 0.01 * iteration)


def _sharded_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import gsx  # noqa: F401
    from gsx import distributed as gdist
    gdist.init_from_env(backend="gloo")
    for N in (1000, 1001):                                          # divisible by the world size / not (uneven blocks)
        shapes = [(N, 3), (N, 16, 3), (N, 3), (N, 4), (N, 1)]
        g0 = torch.Generator().manual_seed(7)
        init = [torch.randn(s, generator=g0) for s in shapes]        # identical replicas
        g = torch.Generator().manual_seed(100 + rank)
        grads = [torch.randn(s, generator=g) for s in shapes]        # this rank's camera
        out = {}
        for mode in ("replicated", "sharded"):
            params = [t.clone().requires_grad_(True) for t in init]
            bucket = gdist.GradBucket(params)
            for p, gr in zip(params, grads):
                p.grad.copy_(gr)
            opt = _RowSGD(params)
            if mode == "replicated":
                h = bucket.all_reduce_mean(async_op=True)            # the asynchronous handle completes the MEAN
                h.wait()
                opt.step(3)
            else:
                gdist.ShardedAdam(opt, bucket).step(3)
                per = (N + world - 1) // world
                assert opt.calls == [(min(N, rank * per), min(N, (rank + 1) * per))]
            out[mode] = np.concatenate([p.detach().numpy().reshape(-1) for p in params])
        assert np.array_equal(out["replicated"], out["sharded"])
        np.save(os.path.join(out_dir, "sharded_%d_%d.npy" % (N, rank)), out["sharded"])
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_optimizer_step_equals_replicated(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_sharded_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for N in (1000, 1001):
        assert np.array_equal(np.load(tmp_path / ("sharded_%d_0.npy" % N)), np.load(tmp_path / ("sharded_%d_1.npy" % N)))


# ---- colour-gradient exchange: 3 floats per (camera, Gaussian) travel, every rank runs the SH backward over all cameras -----------
def _sh_colors_bwd_cpu(sh_degree, means, viewmats, sh, radii, colors, v_colors, v_means_in, v_coeffs_out, v_means_out):
    """CPU stand-in for ops.sh_colors_bwd in its pre-masked mode (radii = colors = None), on the oracle's SH backward."""
    from oracle import oracle
    assert radii is None and colors is None and v_means_in is None
    C, N = viewmats.shape[0], means.shape[0]
    vc = np.zeros(tuple(sh.shape), np.float32)
    vm = np.zeros((N, 3), np.float32)
    for c in range(C):
        campos = np.linalg.inv(viewmats[c].numpy().astype(np.float64))[:3, 3].astype(np.float32)
        dirs = means.detach().numpy() - campos
        a, b = oracle.sh_bwd(sh_degree, dirs, sh.detach().numpy(), None, v_colors[c].numpy(), True)
        vc += a
        vm += b
    v_coeffs_out.copy_(torch.from_numpy(vc))
    v_means_out.copy_(torch.from_numpy(vm))
    return v_coeffs_out, v_means_out


def _xch_inputs(rank, N=257, K=16):
    g = torch.Generator().manual_seed(100 + rank)
    shared = torch.Generator().manual_seed(7)
    means = torch.randn(N, 3, generator=shared) + torch.tensor([0.0, 0.0, 5.0])
    sh = (torch.rand(N, K, 3, generator=shared) - 0.5) * 0.3
    vm = torch.eye(4)
    vm[0, 3] = 0.3 * (rank + 1)
    vm[1, 3] = -0.2 * rank
    colors = torch.rand(1, N, 3, generator=g) - 0.2            # some channels clamped (<= 0)
    colors[0, ::7] = 0.0                                       # Gaussians this camera does not see: zero colour row ...
    v_colors = torch.randn(1, N, 3, generator=g)
    v_colors[0, ::7] = 0.0                                     # ... and zero gradient row
    others = [torch.randn(N, 3, generator=g), torch.randn(N, 3, generator=g), torch.randn(N, 4, generator=g), torch.randn(N, 1, generator=g)]
    return means, sh, vm, colors, v_colors, others   # others: blend's v_means, scaling, rotation, opacity gradients


def _worker_xch(rank, world, port, out_dir, sh_first):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import gsx  # noqa: F401
    from gsx import distributed as gdist
    gdist.init_from_env(backend="gloo")
    means, sh, vm, colors, v_colors, others = _xch_inputs(rank)
    names = ["means", "sh", "scaling_raw", "rotation_raw", "opacity_raw"]
    shapes = dict(means=(257, 3), sh=(257, 16, 3), scaling_raw=(257, 3), rotation_raw=(257, 4), opacity_raw=(257, 1))
    if sh_first:
        names = ["sh", "means", "scaling_raw", "rotation_raw", "opacity_raw"]
    params = [torch.zeros(shapes[n], requires_grad=True) for n in names]
    bucket = gdist.GradBucket(params)
    sinks = bucket.sinks(tuple(names))
    xch = gdist.ColorGradExchange(bucket, names, sh_bwd_fn=_sh_colors_bwd_cpu)
    assert len(xch._runs) == (1 if sh_first else 2)
    vms = [_xch_inputs(r)[2] for r in range(world)]
    xch.begin_step(torch.stack(vms))
    # what the render backward does: the other gradients into their sinks, then the exchange in place of the local SH backward
    sinks["scaling_raw"].copy_(others[1]); sinks["rotation_raw"].copy_(others[2]); sinks["opacity_raw"].copy_(others[3])
    xch.sh_backward(3, means, sh, colors, v_colors, others[0], sinks["sh"], sinks["means"])
    xch.finish()
    np.save(os.path.join(out_dir, "xch_%d.npy" % rank), np.concatenate([sinks[n].numpy().reshape(-1) for n in ["means", "sh", "scaling_raw", "rotation_raw", "opacity_raw"]]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("sh_first,world", [(False, 2), (True, 2), (True, 8)])
def test_color_grad_exchange_equals_dense_all_reduce(tmp_path, sh_first, world):
    """world = 8: the node `north_star` names (one camera per rank on 8 GPUs), over gloo on the CPU."""
    port = _free_port()
    mp.spawn(_worker_xch, args=(world, port, str(tmp_path), sh_first), nprocs=world, join=True)
    b0, b1 = np.load(tmp_path / "xch_0.npy"), np.load(tmp_path / ("xch_%d.npy" % (world - 1)))
    for r in range(1, world):
        assert np.array_equal(b0, np.load(tmp_path / ("xch_%d.npy" % r)))   # bit-identical replicas: same gathered bits, same camera order
    # expectation: every rank's own full gradient (local SH backward of its camera, clamp mask applied), averaged
    exp = None
    for r in range(world):
        means, sh, vm, colors, v_colors, others = _xch_inputs(r)
        vcm = (v_colors * (colors > 0))[0]
        vsh, vmn = torch.zeros(257, 16, 3), torch.zeros(257, 3)
        _sh_colors_bwd_cpu(3, means, vm[None], sh, None, None, vcm[None], None, vsh, vmn)
        full = np.concatenate([(others[0] + vmn).numpy().reshape(-1), vsh.numpy().reshape(-1), others[1].numpy().reshape(-1),
                               others[2].numpy().reshape(-1), others[3].numpy().reshape(-1)])
        exp = full if exp is None else exp + full
    exp = exp / world
    np.testing.assert_allclose(b0, exp, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("sh_first", [False, True])
def test_camera_batch_accumulator_equals_mean_of_single_camera_gradients(sh_first):
    """distributed.CameraBatchAccumulator (C cameras per optimizer step on ONE GPU: BASELINE configs[3]'s iteration without the ranks): after
    C backwards the bucket holds the mean over the cameras of each camera's full gradient — what C ranks x 1 camera produce through
    ColorGradExchange / the dense all-reduce."""
    import gsx  # noqa: F401
    from gsx import distributed as gdist
    C = 3
    names = ["sh", "means", "scaling_raw", "rotation_raw", "opacity_raw"] if sh_first else ["means", "sh", "scaling_raw", "rotation_raw", "opacity_raw"]
    shapes = dict(means=(257, 3), sh=(257, 16, 3), scaling_raw=(257, 3), rotation_raw=(257, 4), opacity_raw=(257, 1))
    params = [torch.zeros(shapes[n], requires_grad=True) for n in names]
    bucket = gdist.GradBucket(params)
    sinks = bucket.sinks(tuple(names))
    acc = gdist.CameraBatchAccumulator(bucket, names, cameras=C, sh_bwd_fn=_sh_colors_bwd_cpu)
    assert len(acc._spans) == (1 if sh_first else 2)
    for _step in range(2):   # a second step on the same object starts from scratch
        acc.begin_step(torch.stack([_xch_inputs(c)[2] for c in range(C)]))
        with pytest.raises(AssertionError):
            acc.finish()
        exp = None
        for c in range(C):
            means, sh, vm, colors, v_colors, others = _xch_inputs(c)
            # what the render backward of camera c does: the other gradients into their sinks (overwriting camera c-1's), then the hook
            sinks["scaling_raw"].copy_(others[1]); sinks["rotation_raw"].copy_(others[2]); sinks["opacity_raw"].copy_(others[3])
            acc.sh_backward(3, means, sh, colors, v_colors, others[0], sinks["sh"], sinks["means"])
            vcm = (v_colors * (colors > 0))[0]
            vsh, vmn = torch.zeros(257, 16, 3), torch.zeros(257, 3)
            _sh_colors_bwd_cpu(3, means, vm[None], sh, None, None, vcm[None], None, vsh, vmn)
            full = np.concatenate([(others[0] + vmn).numpy().reshape(-1), vsh.numpy().reshape(-1), others[1].numpy().reshape(-1),
                                   others[2].numpy().reshape(-1), others[3].numpy().reshape(-1)])
            exp = full if exp is None else exp + full
        acc.finish()
        got = np.concatenate([sinks[n].numpy().reshape(-1) for n in ["means", "sh", "scaling_raw", "rotation_raw", "opacity_raw"]])
        np.testing.assert_allclose(got, exp / C, rtol=2e-5, atol=2e-6)


def _agree_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import gsx  # noqa: F401
    from gsx import distributed as gdist
    gdist.init_from_env(backend="gloo")
    agree = gdist.ListsAgreement()
    # iteration 1: every rank's lists were complete; iteration 2: rank 1 overflowed -> every rank repeats; iteration 3: all complete again
    got = [agree(True), agree(rank != 1), agree(True)]
    # ShardedAdam: a same-size re-index of the optimizer state without merge_moments() must be refused
    p = torch.zeros(8, 3, requires_grad=True)

    class _Opt:   # the two members ShardedAdam.step reads before it touches a collective
        reindex_generation = 0
        state = {}
    sh = gdist.ShardedAdam(_Opt())
    sh._bounds, sh._gen = (8, 0, 4), 0
    _Opt.reindex_generation = 1
    refused = False
    try:
        sh.step(1, gdist.GradBucket([p]))
    except RuntimeError as e:
        refused = "re-indexed" in str(e)
    np.save(os.path.join(out_dir, "agree_%d.npy" % rank), np.array(got + [refused, agree.disagreements, agree.transport == "shm"], dtype=np.int64))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,transport", [(2, "shm"), (2, "gloo"), (8, "shm")])
def test_lists_agreement_and_reindex_guard(tmp_path, monkeypatch, world, transport):
    """distributed.ListsAgreement: the guarded intersection lists of an N-rank step are repeated on EVERY rank when ANY rank overflowed (the
    verdicts meet in a shared-memory page, or in a MIN all-reduce over gloo); ShardedAdam refuses a re-indexed optimizer state whose moments
    were not merged first."""
    monkeypatch.setenv("GSX_AGREE", transport)
    mp.spawn(_agree_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    a, b = np.load(tmp_path / "agree_0.npy"), np.load(tmp_path / "agree_1.npy")
    assert int(a[5]) == (1 if transport == "shm" else 0)   # the transport that was asked for ran
    for r in range(2, world):
        assert np.load(tmp_path / ("agree_%d.npy" % r))[:3].tolist() == [1, 0, 1]
    assert a[:3].tolist() == [1, 0, 1] and b[:3].tolist() == [1, 0, 1]   # both ranks repeat iteration 2
    assert a[3] == 1 and b[3] == 1                                       # the guard fired on both
    assert a[4] == 1 and b[4] == 0                                       # rank 0 repeated because of the OTHER rank


def _single_rank_worker(_, port, out_dir):
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        os.environ.pop(k, None)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GSX_SINGLE_RANK_GROUP="1")
    import gsx  # noqa: F401
    from gsx import distributed as gdist
    assert not gdist.active()
    r, lr, w = gdist.init_from_env(backend="gloo")
    assert (r, w) == (0, 1) and gdist.SINGLE_RANK_COLLECTIVES and gdist.active() and dist.get_world_size() == 1
    g = torch.Generator().manual_seed(3)
    params = [torch.zeros(40, 3, requires_grad=True), torch.zeros(40, 16, 3, requires_grad=True), torch.zeros(40, requires_grad=True)]
    bucket = gdist.GradBucket(params)
    vals = [torch.randn(p.shape, generator=g) for p in params]
    for p, v in zip(params, vals):
        p.grad.copy_(v)
    bucket.last_reduced_bytes = 0
    bucket.all_reduce_mean()                       # executed (a group of one rank would normally skip it): the mean over one rank
    assert bucket.last_reduced_bytes == bucket.nbytes()
    h = bucket.all_reduce_mean_tail_async(1)
    assert h is not None
    bucket.all_reduce_mean_head(h)
    vis = torch.zeros(40, dtype=torch.bool)
    vis[::3] = True
    for p in params:                               # rows no camera sees are zero, as after a render backward
        p.grad[~vis] = 0
    expect = [p.grad.clone() for p in params]
    bucket.all_reduce_mean_rows(vis)
    assert bucket.last_reduced_bytes < bucket.nbytes()
    for p, e in zip(params, expect):
        assert torch.equal(p.grad, e)
    agree = gdist.ListsAgreement()
    assert agree.group is not None and agree(True) is True and agree(False) is False
    open(os.path.join(out_dir, "single_ok"), "w").write("ok")
    dist.destroy_process_group()


def test_single_rank_group_executes_the_collectives(tmp_path, capfd):
    """GSX_SINGLE_RANK_GROUP=1: a process group of ONE rank whose collectives run all the same (the diagnostic that puts the N > 1 code
    path through RCCL on a 1-GPU box; here over gloo) — and gloo's mesh report does not land on stdout."""
    mp.spawn(_single_rank_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    assert (tmp_path / "single_ok").read_text() == "ok"
    out = capfd.readouterr().out
    assert "[Gloo]" not in out, out
