"""tools/dump_reference_cuda.cpp + tools/export_scene.py + tests/golden/ref_dump.py: the route by which a CUDA box pins UT
projection / intersect_offset / blend against the reference's own kernels.  Here the tool is built against THIS backend
(compat/gsplat + libgsx_gsplat_backend.so): CPU tests cover the file format and that the tool compiles and links by the reference's
header names; the GPU test runs it and checks the self-dump against the CPU oracle.  Dumps made on a CUDA box
(tests/golden/ref_cuda/<name>/{scene,dump} or $GSX_REF_CUDA_DUMPS) are compared with north_star's tolerances when present."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.golden import ref_dump
from tests.helpers import rel_l2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gaussian-splatting-cuda_amd")
SRC = os.path.join(ROOT, "tools", "dump_reference_cuda.cpp")
EXE = os.path.join(ROOT, "tools", "dump_reference_cuda")


def _build():
    sys.path.insert(0, PKG)
    try:
        import build as gbuild
    finally:
        sys.path.pop(0)
    gbuild.build_all()
    if os.path.exists(EXE) and os.path.getmtime(EXE) >= max(os.path.getmtime(SRC), os.path.getmtime(gbuild.BACKEND)):
        return EXE
    inc, link = gbuild.torch_cxx_flags()
    cmd = ["g++", "-O1", "-I" + os.path.join(ROOT, "compat", "gsplat")] + inc + [SRC, "-o", EXE, "-L" + PKG, "-lgsx_gsplat_backend", "-lgsx"] + link + \
          ["-Wl,-rpath," + PKG]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-4000:]
    return EXE


def _small_scene():
    import gsx  # noqa: F401
    from gsx import scenes
    sc = scenes.scene_small(seed=42, N=4000)
    sc["width"], sc["height"] = 128, 96
    sc["K"] = scenes.intrinsics(100.0, 100.0, 64.0, 48.0)
    return sc


def test_dump_format_round_trip(tmp_path):
    t = {"a": np.arange(12, dtype=np.float32).reshape(3, 4), "b": np.array([[1, -2]], np.int32), "c": np.array([2 ** 40], np.int64),
         "m": np.array([True, False, True])}
    ref_dump.write_dir(str(tmp_path), t)
    back = ref_dump.read_dir(str(tmp_path))
    assert set(back) == set(t)
    for k in ("a", "b", "c"):
        assert back[k].dtype == t[k].dtype and np.array_equal(back[k], t[k])
    assert back["m"].dtype == np.uint8 and np.array_equal(back["m"], [1, 0, 1])


def test_dump_tool_builds_and_exporter_writes_a_scene(tmp_path):
    exe = _build()
    assert os.access(exe, os.X_OK)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import export_scene
    finally:
        sys.path.pop(0)
    sc = _small_scene()
    export_scene.export(sc, str(tmp_path / "scene"))
    back = ref_dump.read_dir(str(tmp_path / "scene"))
    assert np.array_equal(back["means"], sc["means"].numpy()) and list(back["dims"]) == [128, 96, 0]
    assert back["v_render_colors"].shape == (1, 96, 128, 3)
    r = subprocess.run([exe], stdout=subprocess.PIPE, text=True)   # no arguments: usage, exit code 1 (no GPU touched)
    assert r.returncode == 1 and "usage" in r.stdout


def _compare_dump_with_oracle(scene_t, dump, tag):
    """dump = output of the tool; the oracle is run on the dump's OWN projection / binning so that each stage is compared on
    identical inputs (stage-wise parity)."""
    from oracle import oracle
    f32 = lambda k: np.ascontiguousarray(scene_t[k], np.float32)  # noqa: E731
    W, H, deg = (int(x) for x in scene_t["dims"])
    means, quats, scales, opac = f32("means"), f32("quats"), f32("scales"), f32("opacities")
    vm, K, bg = f32("viewmat").reshape(1, 4, 4), f32("K").reshape(1, 3, 3), f32("background").reshape(1, 3)
    radii, means2d, depths, conics, _ = oracle.projection_ut(means, quats, scales, opac, vm, K, W, H)
    valid_o, valid_d = (radii > 0).all(-1), (dump["radii"] > 0).all(-1)
    both = valid_o & valid_d
    n = means.shape[0]
    assert (valid_o != valid_d).sum() <= max(2, 1e-4 * n), tag                       # cull decisions
    assert (np.abs(radii - dump["radii"])[both].max() <= 1) and ((radii != dump["radii"])[both].mean() < 5e-3), tag
    assert np.abs(means2d - dump["means2d"])[both].max() < 2e-3, tag                  # pixels
    assert np.abs(depths - dump["depths"])[both].max() < 1e-5 * np.abs(depths[both]).max(), tag
    # intersection of the dump's own projection: exact
    tw, th = (W + 15) // 16, (H + 15) // 16
    tpg, ids, fl = oracle.intersect_tile(dump["means2d"], dump["radii"], dump["depths"], 1, 16, tw, th, True)
    assert np.array_equal(tpg, dump["tiles_per_gauss"]) and np.array_equal(ids, dump["isect_ids"]) and np.array_equal(fl, dump["flatten_ids"]), tag
    off = oracle.intersect_offset(ids, 1, tw, th)
    assert np.array_equal(off, dump["isect_offsets"]), tag
    # blend on the dump's colours and binning
    ren, alp, last, frag = oracle.rasterize_fwd(means, quats, scales, dump["colors"], opac[None], bg, None, W, H, 16, vm, K, off, fl, frag_rel=1e-4)
    ok = frag == 0
    err = np.abs(ren - dump["renders"])
    assert ok.mean() > 0.99 and err[ok].max() < 1e-4, (tag, float(ok.mean()), float(err[ok].max()))   # north_star: 1e-4 RGB L-inf
    assert err.max() < dump["colors"].max() / 255.0 + 2e-4, tag
    assert np.abs(alp - dump["alphas"])[ok].max() < 1e-4 and np.array_equal(last[ok], dump["last_ids"][ok]), tag
    g = oracle.rasterize_bwd(means, quats, scales, dump["colors"], opac[None], bg, None, W, H, 16, vm, K, off, fl, dump["alphas"], dump["last_ids"],
                             scene_t["v_render_colors"], scene_t["v_render_alphas"])
    for name, r in zip(["v_means", "v_quats", "v_scales", "v_colors", "v_opacities"], g):
        assert rel_l2(dump[name], r) < 1e-3, (tag, name, rel_l2(dump[name], r))       # north_star: 1e-3 gradient rel-L2


@pytest.mark.gpu
def test_dump_tool_end_to_end_against_oracle(tmp_path):
    exe = _build()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import export_scene
    finally:
        sys.path.pop(0)
    scene_t = export_scene.export(_small_scene(), str(tmp_path / "scene"))
    os.makedirs(tmp_path / "dump")
    r = subprocess.run([exe, str(tmp_path / "scene"), str(tmp_path / "dump")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "dumped" in r.stdout, r.stdout[-2000:]
    dump = ref_dump.read_dir(str(tmp_path / "dump"))
    assert len(dump) == 19
    _compare_dump_with_oracle(scene_t, dump, "self-dump")


def _cuda_dump_dirs():
    roots = [os.path.join(ROOT, "tests", "golden", "ref_cuda")] + ([os.environ["GSX_REF_CUDA_DUMPS"]] if os.environ.get("GSX_REF_CUDA_DUMPS") else [])
    return sorted(d for r in roots for d in glob.glob(os.path.join(r, "*")) if os.path.isdir(os.path.join(d, "scene")) and os.path.isdir(os.path.join(d, "dump")))


@pytest.mark.skipif(not _cuda_dump_dirs(), reason="no reference CUDA dump present (tools/dump_reference_cuda.md)")
def test_oracle_against_reference_cuda_dumps():
    """The pin: the CPU oracle against tensors produced by the reference's own CUDA kernels."""
    for d in _cuda_dump_dirs():
        _compare_dump_with_oracle(ref_dump.read_dir(os.path.join(d, "scene")), ref_dump.read_dir(os.path.join(d, "dump")), os.path.basename(d))
