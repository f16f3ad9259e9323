"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE'S OWN CODE
(tests/torch_impl.cpp compiled unmodified into oracle/_ref by oracle/build_ref.sh).

Run in the build container (needs /root/reference):  python tests/golden/gen_golden.py
Inputs mirror the reference's pinned tests:
  * SH fwd + grads, degrees 0-4, K=25  — tests/test_numerical_gradients.cpp:158-229
  * intersect_tile C=3, N=1000, 40x60, tile 16 — tests/test_garden_data.cpp:531-570
  * quat -> rotmat — tests/torch_impl.cpp:8-35
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    assert ref.available(), "build oracle/_ref first: bash oracle/build_ref.sh"
    rng = np.random.default_rng(42)

    # --- spherical harmonics ---
    N, K = 160, 25
    coeffs = rng.standard_normal((N, K, 3)).astype(np.float32)
    dirs = rng.standard_normal((N, 3)).astype(np.float32)
    v_colors = rng.standard_normal((N, 3)).astype(np.float32)
    sh = {"coeffs": coeffs, "dirs": dirs, "v_colors": v_colors}
    for deg in range(5):
        colors, v_coeffs, v_dirs = ref.spherical_harmonics(deg, dirs, coeffs, v_colors)
        sh[f"colors_{deg}"] = colors
        sh[f"v_coeffs_{deg}"] = v_coeffs
        sh[f"v_dirs_{deg}"] = v_dirs
    np.savez_compressed(os.path.join(OUT, "sh_torch_impl.npz"), **sh)

    # --- tile intersection (the reference test's shapes) ---
    C, Ng, W, H, T = 3, 1000, 40, 60, 16
    tw, th = (W + T - 1) // T, (H + T - 1) // T
    means2d = (rng.standard_normal((C, Ng, 2)) * W).astype(np.float32)
    radii = rng.integers(0, W, (C, Ng, 2)).astype(np.int32)
    depths = rng.random((C, Ng)).astype(np.float32)
    tpg, ids, fl = ref.isect_tiles(means2d, radii, depths, T, tw, th, True)
    np.savez_compressed(os.path.join(OUT, "isect_torch_impl.npz"), means2d=means2d, radii=radii, depths=depths,
                        tile_size=T, tile_width=tw, tile_height=th, tiles_per_gauss=tpg, isect_ids=ids, flatten_ids=fl)
    # second case: single camera, 256x256 (cam id 0 => key encodings agree for any tile count)
    C, Ng, W, H = 1, 2000, 256, 256
    tw, th = (W + T - 1) // T, (H + T - 1) // T
    means2d = (rng.random((C, Ng, 2)) * np.array([W, H]) * 1.2 - 0.1 * W).astype(np.float32)
    radii = rng.integers(0, 24, (C, Ng, 2)).astype(np.int32)
    depths = (rng.random((C, Ng)) * 10 + 0.01).astype(np.float32)
    tpg, ids, fl = ref.isect_tiles(means2d, radii, depths, T, tw, th, True)
    np.savez_compressed(os.path.join(OUT, "isect_torch_impl_256.npz"), means2d=means2d, radii=radii, depths=depths,
                        tile_size=T, tile_width=tw, tile_height=th, tiles_per_gauss=tpg, isect_ids=ids, flatten_ids=fl)

    # --- quaternion -> rotation matrix ---
    quats = rng.standard_normal((256, 4)).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "quat_torch_impl.npz"), quats=quats, rotmats=ref.quat_to_rotmat(quats))
    # covariance / precision of (quat, scale): pins the M = diag(1/s) R^T factor the world-space blend evaluates Gaussians with
    scales = np.exp(rng.uniform(-3.0, 0.5, (quats.shape[0], 3))).astype(np.float32)
    cov, pre = ref.quat_scale_to_covar_preci(quats, scales)
    np.savez_compressed(os.path.join(OUT, "covar_preci_torch_impl.npz"), quats=quats, scales=scales, covars=cov, precis=pre)
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
