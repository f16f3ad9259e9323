"""TEST INFRASTRUCTURE ONLY — loader of oracle/_ref/gsplat_ref_strategy.so: the reference's OWN training host logic (gs::training::MCMC of
src/training/strategies/mcmc.cpp, strategy_utils.cpp, optimizers/fused_adam.cpp, scheduler.cpp — compiled unmodified by
oracle/build_ref_strategy.sh) linked against this repository's drop-in backend.  Only tests/ may import this module.

    m = mod.RefMCMC(sh_degree, active_sh_degree, means, sh0, shN, scaling_raw, rotation_raw, opacity_raw, scene_scale, {param: value, ...})
    m.set_grads([6 tensors]); m.post_backward(it); m.step(it); m.params() -> 6 tensors; m.state(group) -> (exp_avg, exp_avg_sq, step_count);
    m.lr(group); m.size(); m.active_sh_degree(); m.is_refining(it)
Random draws (torch::multinomial, torch::randn_like) come from the process's default CUDA generator: torch.cuda.manual_seed(s) seeds them."""
import importlib.util
import os

HERE = os.path.dirname(os.path.abspath(__file__))
_cache = {}


def path():
    return os.path.join(HERE, "_ref", "gsplat_ref_strategy.so")


def load():
    """The extension module, or None when it has not been built (no /root/reference at build time)."""
    if "m" in _cache:
        return _cache["m"]
    mod = None
    if os.path.exists(path()):
        import torch  # noqa: F401  (libtorch / libamdhip64 first)
        spec = importlib.util.spec_from_file_location("gsplat_ref_strategy", path())
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    _cache["m"] = mod
    return mod
