"""Optimization hyper-parameters of the training loop: the reference's `gs::param::OptimizationParameters`
(include/core/parameters.hpp:15-93) with its JSON (de)serialisation (src/core/parameters.cpp:233-470) and the two parameter
files it ships — `parameter/default_optimization_params.json` (the file BASELINE configs[2] names) and
`parameter/mcmc_optimization_params.json`.

Note the three sources disagree, upstream too: the C++ struct defaults (parameters.hpp:17-37: means_lr 1.6e-4, stop_refine 25000,
regularisers 0.01, strategy "mcmc") are what a run gets when NO file is read; the JSON files carry their own values (default file:
means_lr 1.6e-5, stop_refine 15000, regularisers 0, strategy "default").  `OptimizationParameters()` = the struct defaults,
`OptimizationParameters.from_file(...)` / `.preset("default" | "mcmc")` = a parameter file.
"""
import dataclasses
import json
from dataclasses import dataclass, field
from typing import List

RENDER_MODES = ("RGB", "D", "ED", "RGB_D", "RGB_ED")   # parameters.cpp:330-339
POSE_OPTIMIZATIONS = ("none", "direct", "mlp")          # :341-348
STRATEGIES = ("mcmc", "default")                        # :350-357
# keys from_json reads unconditionally (parameters.cpp:296-308): a file without them is an error
REQUIRED = ("iterations", "means_lr", "shs_lr", "opacity_lr", "scaling_lr", "rotation_lr", "lambda_dssim", "min_opacity", "refine_every",
            "start_refine", "stop_refine", "grad_threshold", "sh_degree")
# JSON key -> field name where they differ (parameters.cpp:262, :379)
_RENAMED = {"skip_intermediate": "skip_intermediate_saving"}


@dataclass
class OptimizationParameters:
    """Field defaults = include/core/parameters.hpp:17-93 (what the reference uses without a parameter file)."""
    iterations: int = 30000
    sh_degree_interval: int = 1000
    means_lr: float = 0.00016
    shs_lr: float = 0.0025
    opacity_lr: float = 0.05
    scaling_lr: float = 0.005
    rotation_lr: float = 0.001
    lambda_dssim: float = 0.2
    min_opacity: float = 0.005
    refine_every: int = 100
    start_refine: int = 500
    stop_refine: int = 25000
    grad_threshold: float = 0.0002
    sh_degree: int = 3
    opacity_reg: float = 0.01
    scale_reg: float = 0.01
    init_opacity: float = 0.5
    init_scaling: float = 0.1
    num_workers: int = 16
    max_cap: int = 1000000
    eval_steps: List[int] = field(default_factory=lambda: [7000, 30000])
    save_steps: List[int] = field(default_factory=lambda: [7000, 30000])
    skip_intermediate_saving: bool = False
    bg_modulation: bool = False
    enable_eval: bool = False
    enable_save_eval_images: bool = True
    render_mode: str = "RGB"
    strategy: str = "mcmc"
    pose_optimization: str = "none"
    use_bilateral_grid: bool = False
    bilateral_grid_X: int = 16
    bilateral_grid_Y: int = 16
    bilateral_grid_W: int = 8
    bilateral_grid_lr: float = 2e-3
    tv_loss_weight: float = 10.0
    prune_opacity: float = 0.005
    grow_scale3d: float = 0.01
    grow_scale2d: float = 0.05
    prune_scale3d: float = 0.1
    prune_scale2d: float = 0.15
    reset_every: int = 3000
    pause_refine_after_reset: int = 0
    revised_opacity: bool = False
    gut: bool = False
    steps_scaler: float = 0.0
    antialiasing: bool = False
    random: bool = False
    init_num_pts: int = 100000
    init_extent: float = 3.0
    save_sog: bool = False
    sog_iterations: int = 10
    enable_sparsity: bool = False
    sparsify_steps: int = 15000
    init_rho: float = 0.0005
    prune_ratio: float = 0.6

    # ---- parameters.cpp:293-470 ------------------------------------------------------------------------------------------
    @classmethod
    def from_json(cls, data: dict) -> "OptimizationParameters":
        missing = [k for k in REQUIRED if k not in data]
        if missing:
            raise KeyError("optimization parameter file lacks required key(s): " + ", ".join(missing))
        p = cls()
        names = {f.name for f in dataclasses.fields(cls)}
        for key, value in data.items():
            name = _RENAMED.get(key, key)
            if name not in names:
                continue  # unknown keys are reported upstream (verify_optimization_parameters) but do not fail the load
            cur = getattr(p, name)
            if isinstance(cur, bool):
                value = bool(value)
            elif isinstance(cur, int):
                value = int(value)
            elif isinstance(cur, float):
                value = float(value)
            elif isinstance(cur, list):
                value = [int(v) for v in value]
            setattr(p, name, value)
        if p.render_mode not in RENDER_MODES:
            raise ValueError("Invalid render mode '%s'. Valid modes are: %s" % (p.render_mode, ", ".join(RENDER_MODES)))
        if p.pose_optimization not in POSE_OPTIMIZATIONS:
            raise ValueError("Invalid pose optimization '%s'. Valid values are: %s" % (p.pose_optimization, ", ".join(POSE_OPTIMIZATIONS)))
        if p.strategy not in STRATEGIES:
            raise ValueError("Invalid optimization strategy '%s'. Valid strategies are: %s" % (p.strategy, ", ".join(STRATEGIES)))
        return p

    def to_json(self) -> dict:
        """parameters.cpp:233-291: every field except the CLI-only ones (gut, headless, rc, preload_to_ram), with upstream's key names."""
        d = dataclasses.asdict(self)
        d.pop("gut")
        d["skip_intermediate"] = d.pop("skip_intermediate_saving")
        return d

    @classmethod
    def from_file(cls, path) -> "OptimizationParameters":
        with open(path) as f:
            return cls.from_json(json.load(f))

    @classmethod
    def preset(cls, name: str) -> "OptimizationParameters":
        """The parameter files the reference ships: "default" = parameter/default_optimization_params.json:1-48 (BASELINE configs[2]),
        "mcmc" = parameter/mcmc_optimization_params.json:1-49."""
        return cls.from_json(PRESETS[name])


# parameter/default_optimization_params.json:1-48, value for value
_DEFAULT_FILE = {
    "iterations": 30000, "sh_degree_interval": 1000, "means_lr": 0.000016, "shs_lr": 0.0025, "opacity_lr": 0.05, "scaling_lr": 0.005,
    "rotation_lr": 0.001, "lambda_dssim": 0.2, "min_opacity": 0.005, "refine_every": 100, "start_refine": 500, "stop_refine": 15000,
    "grad_threshold": 0.0002, "sh_degree": 3, "opacity_reg": 0.0, "scale_reg": 0.0, "init_opacity": 0.1, "init_scaling": 1.0,
    "max_cap": 1000000, "render_mode": "RGB", "strategy": "default", "eval_steps": [7000, 30000], "save_steps": [7000, 30000],
    "enable_eval": False, "enable_save_eval_images": True, "use_bilateral_grid": False, "skip_intermediate": False, "bg_modulation": False,
    "bilateral_grid_X": 16, "bilateral_grid_Y": 16, "bilateral_grid_W": 8, "bilateral_grid_lr": 0.002, "tv_loss_weight": 5.0,
    "prune_opacity": 0.005, "grow_scale3d": 0.01, "grow_scale2d": 0.05, "prune_scale3d": 0.1, "prune_scale2d": 0.15, "reset_every": 3000,
    "pause_refine_after_reset": 0, "revised_opacity": False, "steps_scaler": 0, "antialiasing": False, "random": False,
    "init_num_pts": 100000, "init_extent": 3.0,
}
# parameter/mcmc_optimization_params.json differs from the default file in exactly these entries
_MCMC_FILE = dict(_DEFAULT_FILE, init_opacity=0.5, init_scaling=0.1, opacity_reg=0.01, scale_reg=0.01, stop_refine=25000, strategy="mcmc",
                  tv_loss_weight=10.0, pose_optimization="none")
PRESETS = {"default": _DEFAULT_FILE, "mcmc": _MCMC_FILE}
