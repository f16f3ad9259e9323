"""Round 6: why the Delta-form needs the INVERSE of the reference's fp32 R_inv, not its transpose (csrc/gsx_record.hpp: make_cam_frame; DESIGN.md section 2, profiles/r06_s8cam_attribution.md).
The reference takes a pose through fp32 quaternions (Cameras.cuh:42-52,258-262: matrix -> quat_cast -> inverse -> mat3_cast): restated here in numpy float32, operation by
operation.  For the S-8cam ring cameras the resulting R_inv is orthonormal only to rounding — 7e-7 on cameras 3 / 5 —, so a Gaussian's camera-space centre taken as
R_inv^T (mu - o) lands up to 7e-6 world units from where the reference's rays (origin o, direction R_inv p) see it; with R_inv^-1 it lands within fp32 rounding of mu - o."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def fp32_frame(vm):
    f = np.float32
    se3 = vm.astype(f).reshape(-1)
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = se3[0], se3[1], se3[2], se3[4], se3[5], se3[6], se3[8], se3[9], se3[10]
    cands = [f(m00 + m11 + m22), f(m00 - m11 - m22), f(m11 - m00 - m22), f(m22 - m00 - m11)]   # glm::quat_cast: w, x, y, z branch by the largest
    bi = int(np.argmax([cands[0]] + [c if c > cands[0] else -np.inf for c in cands[1:]])) if max(cands[1:]) > cands[0] else 0
    big = cands[bi]
    bv = f(np.sqrt(f(big + f(1))) * f(0.5))
    mult = f(f(0.25) / bv)
    q = {0: (bv, (m21 - m12) * mult, (m02 - m20) * mult, (m10 - m01) * mult), 1: ((m21 - m12) * mult, bv, (m10 + m01) * mult, (m02 + m20) * mult),
         2: ((m02 - m20) * mult, (m10 + m01) * mult, bv, (m21 + m12) * mult), 3: ((m10 - m01) * mult, (m02 + m20) * mult, (m21 + m12) * mult, bv)}[bi]
    w, x, y, z = [f(v) for v in q]
    d = f(x * x + y * y + z * z + w * w)
    w, x, y, z = f(w / d), f(-x / d), f(-y / d), f(-z / d)                                   # glm::inverse(quat) = conj / dot
    xx, yy, zz, xz, xy, yz, wx, wy, wz = f(x * x), f(y * y), f(z * z), f(x * z), f(x * y), f(y * z), f(w * x), f(w * y), f(w * z)
    R = np.array([[f(1) - f(2) * f(yy + zz), f(2) * f(xy - wz), f(2) * f(xz + wy)], [f(2) * f(xy + wz), f(1) - f(2) * f(xx + zz), f(2) * f(yz - wx)],
                  [f(2) * f(xz - wy), f(2) * f(yz + wx), f(1) - f(2) * f(xx + yy)]], dtype=f)   # glm::mat3_cast of the NON-normalised quaternion
    return bi, float(d) - 1.0, R


def test_ring_camera_frames_and_the_inverse_the_delta_form_needs():
    scenes = importlib.import_module("gaussian-splatting-cuda_amd.scenes")
    rng = np.random.default_rng(0)
    dm = rng.standard_normal((2000, 3))
    dm = (dm / np.linalg.norm(dm, axis=1, keepdims=True) * 10.0).astype(np.float32)          # Gaussians 10 world units from the camera
    worst_t, worst_i = {}, {}
    for i, vm in enumerate(scenes.ring_cameras(8)):
        bi, qerr, R = fp32_frame(vm.numpy())
        R64 = R.astype(np.float64)
        orth = np.abs(R64 @ R64.T - np.eye(3)).max()
        m_t = (R.T @ dm.T).T                                                                     # rounds 1 - 5: m = R_inv^T (mu - o), fp32
        Ri = np.linalg.inv(R64).astype(np.float32)                                               # round 6: m = R_inv^-1 (mu - o), the inverse formed in double, applied in fp32
        m_i = (Ri @ dm.T).T
        back_t = np.linalg.norm(R64 @ m_t.astype(np.float64).T - dm.astype(np.float64).T, axis=0).max()   # where the reference's rays see the Gaussian vs where the record puts it
        back_i = np.linalg.norm(R64 @ m_i.astype(np.float64).T - dm.astype(np.float64).T, axis=0).max()
        worst_t[i], worst_i[i] = back_t, back_i
        if i in (0, 4):
            assert orth == 0.0 and back_t < 2e-6                  # axis-aligned poses: the round trip is exact
        if i in (3, 5):
            assert bi == 2 and 5e-7 < orth < 1e-6, (bi, orth)      # the y-branch of quat_cast, |q|^2 - 1 = 1.2e-7
            assert back_t > 4e-6, back_t                           # 7e-7 x 10 units: 3.5e-3 sigma of a 0.002 Gaussian, direction dependent
        assert back_i < 2.5e-6, (i, back_i)                        # within fp32 rounding of a length-10 vector on every camera
    assert worst_t[3] > 2.5 * worst_i[3] and worst_t[5] > 2.5 * worst_i[5], (worst_t, worst_i)
