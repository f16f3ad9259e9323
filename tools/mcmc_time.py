"""Times the MCMC strategy's per-iteration host + device work (strategies/mcmc.cpp:114-366 mirrored in gsx/strategy.py) at 1 M and 5 M
Gaussians: inject_noise (every iteration), relocate_gs with 1 % dead Gaussians and add_new_gs (+5 %) (every refine_every iterations).
    python tools/mcmc_time.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import gsx  # noqa: E402,F401
from gsx import parameters, scenes, strategy  # noqa: E402


def bench(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    dev = "cuda:0"
    for name, mk in (("1M", scenes.scene_1m), ("5M", scenes.scene_5m)):
        sc = mk()
        N = sc["means"].shape[0]
        gen = torch.Generator(device=dev)
        gen.manual_seed(0)
        res = {}
        for what in ("inject_noise", "relocate_gs (1% dead)", "add_new_gs (+5%)", "post_backward on a refine iteration"):
            times = []
            for rep in range(3):
                model = scenes.to_splat_data(sc, dev)
                for p in model.params():
                    p.requires_grad_(True)
                    p.grad = torch.zeros_like(p)
                prm = parameters.OptimizationParameters(max_cap=int(N * 1.2))
                mc = strategy.MCMC(model, prm, 1.0, gen)
                mc.optimizer.step(1)
                with torch.no_grad():
                    model.opacity_raw[: N // 100] = -10.0
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                if what == "inject_noise":
                    mc.inject_noise()
                elif what.startswith("relocate"):
                    mc.relocate_gs()
                elif what.startswith("add_new"):
                    mc.add_new_gs()
                else:
                    mc.post_backward(600)
                torch.cuda.synchronize()
                times.append((time.perf_counter() - t0) * 1e3)
                del model, mc
            res[what] = min(times[1:])
        print("MCMC at %s Gaussians: " % name + "  ".join("%s %.2f ms" % kv for kv in res.items()))


if __name__ == "__main__":
    main()
