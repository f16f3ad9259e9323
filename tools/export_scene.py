#!/usr/bin/env python
"""Writes one of the synthetic scenes of SURVEY.md §8d (same seeds as the tests and bench.py: gsx/scenes.py) in the raw format
tools/dump_reference_cuda.cpp reads, including the upstream gradients of the backward (seeded numpy, as the parity tests)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.golden import ref_dump  # noqa: E402


def export(scene, out_dir, grad_seed=0):
    os.makedirs(out_dir, exist_ok=True)
    H, W = scene["height"], scene["width"]
    rng = np.random.default_rng(grad_seed)
    t = {k: scene[k].numpy() for k in ("means", "quats", "scales", "opacities", "sh", "viewmat", "K", "background")}
    t["dims"] = np.array([W, H, scene["sh_degree"]], np.int32)
    t["v_render_colors"] = rng.standard_normal((1, H, W, 3)).astype(np.float32)
    t["v_render_alphas"] = rng.standard_normal((1, H, W, 1)).astype(np.float32)
    ref_dump.write_dir(out_dir, t)
    return t


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="small", choices=["small", "1m", "5m"])
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    import gsx  # noqa: F401
    from gsx import scenes
    sc = {"small": scenes.scene_small, "1m": scenes.scene_1m, "5m": scenes.scene_5m}[a.scene]()
    export(sc, a.out)
    print("wrote", a.out)
