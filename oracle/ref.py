"""ctypes loader for oracle/_ref/libtorchimpl_ref.so — the reference's own tests/torch_impl.cpp
compiled unmodified (oracle/build_ref.sh).  TEST INFRASTRUCTURE ONLY.  Available only where the
library has been built (this container; it travels to the GPU box as a prebuilt .so)."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libtorchimpl_ref.so")
_lib = None


def available():
    return os.path.exists(LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        import torch  # noqa: F401  (makes libtorch resolvable before dlopen)
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.ref_isect_tiles.restype = ctypes.c_int64
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def quat_to_rotmat(quats):
    q = np.ascontiguousarray(quats, np.float32).reshape(-1, 4)
    out = np.empty((q.shape[0], 3, 3), np.float32)
    lib().ref_quat_to_rotmat(ctypes.c_int64(q.shape[0]), _p(q), _p(out))
    return out


def quat_scale_to_covar_preci(quats, scales):
    q = np.ascontiguousarray(quats, np.float32).reshape(-1, 4)
    s = np.ascontiguousarray(scales, np.float32).reshape(-1, 3)
    cov, pre = np.empty((q.shape[0], 3, 3), np.float32), np.empty((q.shape[0], 3, 3), np.float32)
    lib().ref_quat_scale_to_covar_preci(ctypes.c_int64(q.shape[0]), _p(q), _p(s), _p(cov), _p(pre))
    return cov, pre


def spherical_harmonics(degree, dirs, coeffs, v_colors=None):
    d = np.ascontiguousarray(dirs, np.float32)
    c = np.ascontiguousarray(coeffs, np.float32)
    n, K = d.shape[0], c.shape[1]
    colors = np.empty((n, 3), np.float32)
    if v_colors is None:
        lib().ref_spherical_harmonics(ctypes.c_int(degree), ctypes.c_int64(n), ctypes.c_int64(K), _p(d), _p(c),
                                      _p(colors), None, None, None)
        return colors
    v = np.ascontiguousarray(v_colors, np.float32)
    v_coeffs = np.empty((n, K, 3), np.float32)
    v_dirs = np.empty((n, 3), np.float32)
    lib().ref_spherical_harmonics(ctypes.c_int(degree), ctypes.c_int64(n), ctypes.c_int64(K), _p(d), _p(c), _p(colors),
                                  _p(v), _p(v_coeffs), _p(v_dirs))
    return colors, v_coeffs, v_dirs


def isect_tiles(means2d, radii, depths, tile_size, tile_width, tile_height, sort=True):
    m = np.ascontiguousarray(means2d, np.float32)
    r = np.ascontiguousarray(radii, np.int32)
    d = np.ascontiguousarray(depths, np.float32)
    C, N = d.shape
    tpg = np.empty((C, N), np.int32)
    cap = int(C * N * tile_width * tile_height)
    ids = np.empty((cap,), np.int64)
    fl = np.empty((cap,), np.int32)
    n = lib().ref_isect_tiles(ctypes.c_int64(C), ctypes.c_int64(N), _p(m), _p(r), _p(d), ctypes.c_int(tile_size),
                              ctypes.c_int(tile_width), ctypes.c_int(tile_height), ctypes.c_int(int(sort)), _p(tpg),
                              ctypes.c_int64(cap), _p(ids), _p(fl))
    return tpg, ids[:n].copy(), fl[:n].copy()


def ewa_projection(means, quats, scales, viewmats, Ks, width, height, eps2d=0.3, near_plane=0.01, far_plane=1e10):
    m = np.ascontiguousarray(means, np.float32)
    q = np.ascontiguousarray(quats, np.float32)
    s = np.ascontiguousarray(scales, np.float32)
    vm = np.ascontiguousarray(viewmats, np.float32)
    K = np.ascontiguousarray(Ks, np.float32)
    C, N = vm.shape[0], m.shape[0]
    radii = np.empty((C, N, 2), np.int32)
    m2d = np.empty((C, N, 2), np.float32)
    dep = np.empty((C, N), np.float32)
    con = np.empty((C, N, 3), np.float32)
    lib().ref_ewa_projection(ctypes.c_int64(C), ctypes.c_int64(N), _p(m), _p(q), _p(s), _p(vm), _p(K),
                             ctypes.c_int(width), ctypes.c_int(height), ctypes.c_float(eps2d),
                             ctypes.c_float(near_plane), ctypes.c_float(far_plane), _p(radii), _p(m2d), _p(dep), _p(con))
    return radii, m2d, dep, con
