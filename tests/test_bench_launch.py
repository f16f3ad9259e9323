"""bench.py's launch contract: `python bench.py --gpus N` with no launcher around it must start N ranks itself
(torch.distributed.run, rendezvous on 127.0.0.1), and still work when the driver wraps it in torch.distributed.run.  CPU-only: the
--launch-check mode stops after the process group is up (gloo) and one all-reduce."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def _last_json(stdout):
    for line in reversed(stdout.strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise AssertionError("no JSON line in:\n" + stdout[-2000:])


def test_self_launch_command_is_the_contracts_launch_line():
    sys.path.insert(0, ROOT)
    try:
        import bench
    finally:
        sys.path.pop(0)
    cmd = bench.self_launch_command(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], port=29517)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29517"
    assert cmd[-7] == BENCH and cmd[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]


def test_bare_python_bench_gpus_2_launches_two_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--launch-check"], env=_env(), cwd="/tmp",
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    assert _last_json(r.stdout) == {"launch_check": True, "n_gpus": 2, "backend": "gloo"}


def test_under_torch_distributed_run_it_does_not_relaunch():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29541",
           BENCH, "--gpus", "2", "--backend", "gloo", "--launch-check"]
    r = subprocess.run(cmd, env=_env(), cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert _last_json(r.stdout)["n_gpus"] == 2
    # the job's stdout is the ONE JSON line of rank 0: gloo's own "[Gloo] Rank r is connected to ..." (printed on stdout by the library when a
    # group is created, on every rank) is diverted to stderr (gsx.distributed._stdout_to_stderr)
    assert [ln for ln in r.stdout.splitlines() if ln.strip()] == [json.dumps({"launch_check": True, "n_gpus": 2, "backend": "gloo"})], r.stdout


def test_single_gpu_launch_check_needs_no_process_group():
    r = subprocess.run([sys.executable, BENCH, "--launch-check"], env=_env(), cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and _last_json(r.stdout) == {"launch_check": True, "n_gpus": 1, "backend": None}
