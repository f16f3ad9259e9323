"""Hyper-parameter loader (gsx/parameters.py) against the reference's parameter files and JSON rules (src/core/parameters.cpp)."""
import json
import os

import pytest

import gsx  # noqa: F401
from gsx import parameters

REF = "/root/reference/parameter"


def test_struct_defaults_and_presets_differ_like_upstream():
    s, d, m = parameters.OptimizationParameters(), parameters.OptimizationParameters.preset("default"), parameters.OptimizationParameters.preset("mcmc")
    assert (s.means_lr, s.stop_refine, s.scale_reg, s.strategy) == (0.00016, 25000, 0.01, "mcmc")       # parameters.hpp:19,28,32,44
    assert (d.means_lr, d.stop_refine, d.scale_reg, d.opacity_reg, d.strategy) == (0.000016, 15000, 0.0, 0.0, "default")
    assert (m.stop_refine, m.scale_reg, m.opacity_reg, m.init_opacity, m.init_scaling, m.strategy) == (25000, 0.01, 0.01, 0.5, 0.1, "mcmc")
    assert d.tv_loss_weight == 5.0 and m.tv_loss_weight == 10.0 and d.max_cap == m.max_cap == 1000000


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
@pytest.mark.parametrize("name", ["default", "mcmc"])
def test_presets_equal_the_reference_files(name):
    data = json.load(open(os.path.join(REF, name + "_optimization_params.json")))
    assert data == parameters.PRESETS[name]
    assert parameters.OptimizationParameters.from_file(os.path.join(REF, name + "_optimization_params.json")) == parameters.OptimizationParameters.preset(name)


def test_json_round_trip_required_keys_and_validation(tmp_path):
    p = parameters.OptimizationParameters.preset("mcmc")
    p.max_cap, p.eval_steps = 123456, [10, 20]
    path = tmp_path / "p.json"
    path.write_text(json.dumps(p.to_json()))
    q = parameters.OptimizationParameters.from_file(path)
    assert q == p and "skip_intermediate" in p.to_json() and "gut" not in p.to_json()
    bad = dict(parameters.PRESETS["default"])
    del bad["means_lr"]
    with pytest.raises(KeyError, match="means_lr"):
        parameters.OptimizationParameters.from_json(bad)
    for key, val in (("strategy", "adc"), ("render_mode", "RGBD"), ("pose_optimization", "yes")):
        with pytest.raises(ValueError):
            parameters.OptimizationParameters.from_json(dict(parameters.PRESETS["default"], **{key: val}))
    # unknown keys do not fail the load (upstream only reports them)
    assert parameters.OptimizationParameters.from_json(dict(parameters.PRESETS["default"], not_a_parameter=1)).iterations == 30000
    # optional keys fall back to the struct defaults
    minimal = {k: parameters.PRESETS["default"][k] for k in parameters.REQUIRED}
    assert parameters.OptimizationParameters.from_json(minimal).opacity_reg == 0.01
