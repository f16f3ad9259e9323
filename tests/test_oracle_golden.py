"""The oracle is pinned against golden vectors produced by the reference's own tests/torch_impl.cpp
(tests/golden/gen_golden.py) with the reference tests' own tolerances:
SH allclose(1e-4, 1e-4) (tests/test_numerical_gradients.cpp:158-229), intersect exact
(tests/test_garden_data.cpp:531-570), quat->rotmat 1e-5."""
import os

import numpy as np
import pytest

from oracle import oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _lexsorted(ids, fl):
    order = np.lexsort((fl, ids))
    return ids[order], fl[order]


@pytest.mark.parametrize("deg", [0, 1, 2, 3, 4])
def test_sh_fwd_bwd_vs_torch_impl(deg):
    g = np.load(os.path.join(GOLDEN, "sh_torch_impl.npz"))
    colors = oracle.sh_fwd(deg, g["dirs"], g["coeffs"])
    np.testing.assert_allclose(colors, g[f"colors_{deg}"], rtol=1e-4, atol=1e-4)
    v_coeffs, v_dirs = oracle.sh_bwd(deg, g["dirs"], g["coeffs"], None, g["v_colors"], True)
    np.testing.assert_allclose(v_coeffs, g[f"v_coeffs_{deg}"], rtol=1e-4, atol=1e-4)
    if deg > 0:
        np.testing.assert_allclose(v_dirs, g[f"v_dirs_{deg}"], rtol=1e-4, atol=1e-4)
    else:
        assert np.all(v_dirs == 0)


def test_sh_f64_agrees_with_f32():
    g = np.load(os.path.join(GOLDEN, "sh_torch_impl.npz"))
    c32 = oracle.sh_fwd(4, g["dirs"], g["coeffs"])
    c64 = oracle.sh_fwd(4, g["dirs"].astype(np.float64), g["coeffs"].astype(np.float64))
    np.testing.assert_allclose(c32, c64, rtol=1e-5, atol=1e-5)


def test_sh_masks_leave_output_untouched():
    g = np.load(os.path.join(GOLDEN, "sh_torch_impl.npz"))
    masks = np.arange(g["dirs"].shape[0]) % 2 == 0
    colors = oracle.sh_fwd(3, g["dirs"], g["coeffs"], masks)
    assert np.all(colors[~masks] == 0)
    np.testing.assert_allclose(colors[masks], g["colors_3"][masks], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("name", ["isect_torch_impl.npz", "isect_torch_impl_256.npz"])
def test_intersect_tile_exact_vs_torch_impl(name):
    g = np.load(os.path.join(GOLDEN, name))
    C = g["depths"].shape[0]
    tpg, ids, fl = oracle.intersect_tile(g["means2d"], g["radii"], g["depths"], C, int(g["tile_size"]),
                                         int(g["tile_width"]), int(g["tile_height"]), True)
    assert np.array_equal(tpg, g["tiles_per_gauss"])
    assert np.array_equal(ids, g["isect_ids"])  # keys, exact
    # torch::argsort is not guaranteed stable: compare (key, value) pairs as multisets...
    a_ids, a_fl = _lexsorted(ids, fl)
    b_ids, b_fl = _lexsorted(g["isect_ids"], g["flatten_ids"])
    assert np.array_equal(a_ids, b_ids) and np.array_equal(a_fl, b_fl)
    # ...and require OUR order to be the stable one (CUB radix sort semantics: ties keep flatten order)
    same_key = ids[1:] == ids[:-1]
    assert np.all(fl[1:][same_key] > fl[:-1][same_key])
    # where keys are unique the value order must match the reference exactly
    uniq = np.ones(ids.shape[0], bool)
    uniq[1:] &= ~same_key
    uniq[:-1] &= ~same_key
    assert np.array_equal(fl[uniq], g["flatten_ids"][uniq])


def test_intersect_offset_properties():
    g = np.load(os.path.join(GOLDEN, "isect_torch_impl.npz"))
    C, tw, th = g["depths"].shape[0], int(g["tile_width"]), int(g["tile_height"])
    ids = g["isect_ids"]
    off = oracle.intersect_offset(ids, C, tw, th).reshape(-1)
    n_tiles = tw * th
    tile_n_bits = int(n_tiles).bit_length()
    flat = ((ids >> 32) >> tile_n_bits) * n_tiles + ((ids >> 32) & ((1 << tile_n_bits) - 1))
    expect = np.searchsorted(flat, np.arange(C * n_tiles), side="left")
    assert np.array_equal(off, expect.astype(np.int32))
    # empty input -> zeros (gsplat/IntersectTile.cu:268-271)
    assert np.all(oracle.intersect_offset(np.zeros((0,), np.int64), C, tw, th) == 0)


def test_quat_to_rotmat_vs_torch_impl():
    g = np.load(os.path.join(GOLDEN, "quat_torch_impl.npz"))
    R = oracle.quat_to_rotmat(g["quats"])
    np.testing.assert_allclose(R, g["rotmats"], rtol=1e-5, atol=1e-5)


def test_preci_half_vs_torch_impl_covar_preci():
    """The factor M = diag(1/s) R^T the world-space blend evaluates every Gaussian with (gro = M (o - mu), grd = M d) against the
    reference's quat_scale_to_covar_preci (tests/torch_impl.cpp:38-78): M^T M = precision, M^-1 M^-T = covariance."""
    g = np.load(os.path.join(GOLDEN, "covar_preci_torch_impl.npz"))
    for dt, tol in ((np.float32, 2e-4), (np.float64, 2e-5)):
        M = oracle.preci_half(g["quats"].astype(dt), g["scales"].astype(dt))
        preci = np.einsum("nki,nkj->nij", M, M)
        scale = np.abs(g["precis"]).max(axis=(1, 2), keepdims=True)
        assert (np.abs(preci - g["precis"]) / scale).max() < tol
        Minv = np.linalg.inv(M.astype(np.float64))
        covar = np.einsum("nik,njk->nij", Minv, Minv)
        cscale = np.abs(g["covars"]).max(axis=(1, 2), keepdims=True)
        assert (np.abs(covar - g["covars"]) / cscale).max() < tol
