"""How long does the guarded-list agreement of N ranks take?  distributed.ListsAgreement is one MIN all-reduce of a 4-byte flag over a gloo
group (loopback TCP) on the host, once per iteration, between the blend backward and the SH backward of every rank (VERDICT r04 "next" #8).
No GPU needed: `python tools/lists_agree_time.py [world=8] [iterations=2000]` spawns `world` processes on this box and prints the time of
the call as rank 0 sees it (every rank enters at its own pace: the all-reduce also absorbs the skew between the ranks, so two figures are
printed — back-to-back calls, and calls entered after a random 0 - 200 us of per-rank "work")."""
import os
import socket
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(rank, world, port, iters, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    import gsx  # noqa: F401
    from gsx import distributed as gdist
    gdist.init_from_env(backend="gloo")
    agree = gdist.ListsAgreement()
    gen = torch.Generator().manual_seed(rank)
    res = {}
    for label, jitter in (("back_to_back", 0.0), ("after_0_200us_of_rank_work", 200e-6)):
        for _ in range(50):
            agree(True)
        dist.barrier()
        ts = []
        for i in range(iters):
            if jitter:
                t_end = time.perf_counter() + float(torch.rand(1, generator=gen)) * jitter
                while time.perf_counter() < t_end:
                    pass
            t0 = time.perf_counter()
            ok = agree(i % 97 != 0 or rank != world - 1)   # now and then one rank votes "repeat"
            ts.append(time.perf_counter() - t0)
            assert ok == (i % 97 != 0)
        ts.sort()
        res[label] = dict(mean_us=1e6 * sum(ts) / len(ts), p50_us=1e6 * ts[len(ts) // 2], p99_us=1e6 * ts[int(len(ts) * 0.99)], max_us=1e6 * ts[-1])
    if rank == 0:
        import json
        print(json.dumps({"world": world, "iterations": iters, "host_cores": os.cpu_count(), "disagreements_seen_by_rank0": agree.disagreements, **res}))
    dist.barrier()
    dist.destroy_process_group()


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    mp.spawn(_worker, args=(world, port, iters, None), nprocs=world, join=True)


if __name__ == "__main__":
    main()
