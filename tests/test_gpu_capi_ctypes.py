"""The C ABI called the way a non-C++ host would (ctypes, raw device pointers, explicit stream) — no shim, no pybind: the stub of
INTEGRATION.md executed for real.  torch only owns the device memory."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    lib = ctypes.CDLL(os.path.join(ROOT, "gaussian-splatting-cuda_amd", "libgsx.so"))
    lib.gsx_last_error.restype = ctypes.c_char_p
    return lib


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def test_sh_fwd_and_adam_through_ctypes():
    lib = _lib()
    dev = "cuda:0"
    stream = torch.cuda.Stream(device=dev)
    rng = np.random.default_rng(0)
    n, K, deg = 3001, 16, 3
    dirs = rng.standard_normal((n, 3)).astype(np.float32)
    coeffs = rng.standard_normal((n, K, 3)).astype(np.float32)
    D, Cf = torch.from_numpy(dirs).to(dev), torch.from_numpy(coeffs).to(dev)
    out = torch.empty(n, 3, device=dev)
    torch.cuda.synchronize()
    rc = lib.gsx_spherical_harmonics_fwd(ctypes.c_uint32(deg), ctypes.c_uint32(n), ctypes.c_uint32(K), _ptr(D), _ptr(Cf), None, _ptr(out),
                                         ctypes.c_void_p(stream.cuda_stream))
    assert rc == 0, lib.gsx_last_error()
    stream.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), oracle.sh_fwd(deg, dirs, coeffs), rtol=1e-4, atol=1e-4)
    # error path: a degree the coefficient count cannot hold is rejected with a message, nothing is launched
    rc = lib.gsx_spherical_harmonics_fwd(ctypes.c_uint32(4), ctypes.c_uint32(n), ctypes.c_uint32(K), _ptr(D), _ptr(Cf), None, _ptr(out), None)
    assert rc == -1 and len(lib.gsx_last_error()) > 0
    # fused Adam on raw pointers
    p = rng.standard_normal(4096).astype(np.float32)
    g = rng.standard_normal(4096).astype(np.float32)
    P, G = torch.from_numpy(p.copy()).to(dev), torch.from_numpy(g).to(dev)
    M, V = torch.zeros(4096, device=dev), torch.zeros(4096, device=dev)
    torch.cuda.synchronize()
    f32 = ctypes.c_float
    rc = lib.gsx_adam_step(ctypes.c_uint64(1), ctypes.c_uint32(4096), ctypes.c_uint64(4096), ctypes.c_uint64(4096), _ptr(P), _ptr(M), _ptr(V), _ptr(G),
                           f32(1e-2), f32(0.9), f32(0.999), f32(1e-15), f32(1.0 / (1 - 0.9)), f32(1.0 / np.sqrt(1 - 0.999)),
                           ctypes.c_void_p(stream.cuda_stream))
    assert rc == 0, lib.gsx_last_error()
    stream.synchronize()
    rp, rm, rv = oracle.adam_step(p, np.zeros_like(p), np.zeros_like(p), g, 1e-2, 0.9, 0.999, 1e-15, 1)
    np.testing.assert_allclose(P.cpu().numpy(), rp, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(V.cpu().numpy(), rv, rtol=1e-6, atol=1e-12)
