"""Pins the CPU oracle (oracle/gsx_oracle.cpp) against golden tensors produced by the REFERENCE's own kernels — gsplat/*.cu compiled
unmodified for gfx950 (oracle/build_ref_hip.sh) and executed on an MI355X by tests/golden/gen_ref_hip_golden.py; fixtures:
tests/golden/ref_hip/*.npz — for the stages no upstream test pins: projection_ut_3dgs_fused (pinhole, OpenCV-distorted
pinhole, fisheye, two rolling shutters, compensations), intersect_tile / intersect_offset, blend forward, blend backward.
Stage by stage on the reference's own intermediate tensors.  Tolerances: north_star's (1e-4 RGB L-inf, 1e-3 gradient rel-L2); integer
outputs exact."""
import os

import numpy as np
import pytest

from oracle import oracle
from tests.golden import ref_hip_cases
from tests.helpers import rel_l2

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "ref_hip")


def _cases():
    import gsx  # noqa: F401
    from gsx import scenes
    return ref_hip_cases.cases(scenes)


CASE_NAMES = ["pinhole_sh3_comp", "distorted_pinhole", "fisheye", "rolling_top_to_bottom", "rolling_left_to_right"]


def test_fixtures_present():
    for n in CASE_NAMES:
        assert os.path.exists(os.path.join(GOLD, n + ".npz")), "missing reference-kernel golden %s.npz (tests/golden/gen_ref_hip_golden.py)" % n


@pytest.mark.parametrize("name", CASE_NAMES)
def test_oracle_against_reference_kernel_outputs(name):
    sc, cam = _cases()[name]
    g = np.load(os.path.join(GOLD, name + ".npz"))
    f = lambda k: np.ascontiguousarray(sc[k].numpy(), np.float32)  # noqa: E731
    W, H = sc["width"], sc["height"]
    tw, th = (W + 15) // 16, (H + 15) // 16
    okw = dict(camera_model=cam.get("camera_model", oracle.PINHOLE), shutter=cam.get("shutter", oracle.SHUTTER_GLOBAL))
    for k in ("viewmats1", "radial", "tangential", "thin_prism"):
        okw[k] = None if cam.get(k) is None else np.asarray(cam[k], np.float32)
    # ---- projection (ProjectionUT3DGSFused.cu:16-203)
    comp = cam.get("calc_compensations", False)
    radii, m2d, dep, con, cp = oracle.projection_ut(f("means"), f("quats"), f("scales"), f("opacities"), f("viewmat")[None], f("K")[None], W, H,
                                                    calc_compensations=comp, **okw)
    vr, vo = (g["radii"] > 0).all(-1), (radii > 0).all(-1)
    assert np.array_equal(vr, vo), "cull decisions differ from the reference kernel"
    assert vr.mean() > 0.5
    assert np.array_equal(radii[vr], g["radii"][vr]), "radii differ from the reference kernel"
    assert np.abs(m2d - g["means2d"])[vr].max() < 5e-3           # px (the UT's -99/+16.67 weights amplify fp32 rounding ~100x)
    assert (np.abs(dep - g["depths"])[vr] / np.abs(g["depths"][vr])).max() < 1e-6
    assert (np.abs(con - g["conics"])[vr] / (np.abs(g["conics"][vr]).max(-1, keepdims=True))).max() < 5e-3
    if comp:
        assert np.abs(cp - g["compensations"])[vr].max() < 1e-4     # sqrt(det / det_blur): the determinant cancels digits
    # ---- intersection on the reference's projection: exact (IntersectTile.cu:23-114, 206-252)
    tpg, ids, fl = oracle.intersect_tile(g["means2d"], g["radii"], g["depths"], 1, 16, tw, th, True)
    off = oracle.intersect_offset(ids, 1, tw, th)
    assert np.array_equal(tpg, g["tiles_per_gauss"]) and np.array_equal(ids, g["isect_ids"])
    assert np.array_equal(fl, g["flatten_ids"]) and np.array_equal(off, g["tile_offsets"])
    # ---- blend forward on the reference's colours and binning (RasterizeToPixelsFromWorld3DGSFwd.cu:19-279)
    args = (f("means"), f("quats"), f("scales"), g["colors"], f("opacities")[None], f("background")[None], None, W, H, 16, f("viewmat")[None], f("K")[None],
            g["tile_offsets"], g["flatten_ids"])
    ren, alp, last = oracle.rasterize_fwd(*args, **okw)
    assert np.abs(ren - g["renders"]).max() < 1e-4, "blend forward: RGB L-inf vs the reference kernel"
    assert np.abs(alp - g["alphas"]).max() < 1e-4 and np.array_equal(last, g["last_ids"])
    # ---- blend backward (RasterizeToPixelsFromWorld3DGSBwd.cu:16-373) on the reference's alphas / last ids
    v_rc, v_ra = ref_hip_cases.upstream_grads(sc)
    grads = oracle.rasterize_bwd(*args, g["alphas"], g["last_ids"], v_rc, v_ra, **okw)
    for n, v in zip(["v_means", "v_quats", "v_scales", "v_colors", "v_opacities"], grads):
        assert rel_l2(v, g[n]) < 1e-3, (n, rel_l2(v, g[n]))
