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


class _UT(ctypes.Structure):
    _fields_ = [("alpha", ctypes.c_float), ("beta", ctypes.c_float), ("kappa", ctypes.c_float), ("margin", ctypes.c_float), ("all_valid", ctypes.c_int32)]


class _Cams(ctypes.Structure):   # include/gsx.h: gsx_cameras
    _fields_ = [("C", ctypes.c_uint32), ("viewmats0", ctypes.c_void_p), ("viewmats1", ctypes.c_void_p), ("Ks", ctypes.c_void_p),
                ("camera_model", ctypes.c_int32), ("shutter", ctypes.c_int32), ("radial", ctypes.c_void_p), ("tangential", ctypes.c_void_p),
                ("thin_prism", ctypes.c_void_p)]


def test_whole_render_chain_through_the_raw_c_abi_on_a_side_stream():
    """INTEGRATION.md's host protocol executed with ctypes only: projection -> SH colours -> binned intersection (count, ONE host read of the
    pinned word, fill) -> blend forward -> blend backward, every buffer and workspace supplied by the caller, every launch on a
    non-default stream.  Checked against the same chain through the C++ shim (bit-exact integers, same floats) and, for the image,
    against the CPU oracle."""
    import gsx  # noqa: F401
    from gsx import ops, scenes
    from tests.helpers import oracle_pipeline
    lib = _lib()
    assert lib.gsx_abi_version() == 7
    for f in ("gsx_intersect_bin_count_workspace_bytes", "gsx_intersect_bin_fill_workspace_bytes", "gsx_rasterize_fwd_workspace_bytes",
              "gsx_rasterize_bwd_workspace_bytes"):
        getattr(lib, f).restype = ctypes.c_size_t
    dev = "cuda:0"
    sc = scenes.scene_small(seed=4, N=3000)
    W = H = 112
    sc["width"], sc["height"], sc["K"] = W, H, scenes.intrinsics(85.0, 85.0, 56.0, 56.0)
    N, C, tw, th = 3000, 1, 7, 7
    u32, i64, f32, sz = ctypes.c_uint32, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t
    d = lambda k: sc[k].to(dev).contiguous()  # noqa: E731
    means, quats, scales, opac, sh = d("means"), d("quats"), d("scales"), d("opacities"), d("sh")
    vm, K, bg = sc["viewmat"][None].to(dev).contiguous(), sc["K"][None].to(dev).contiguous(), sc["background"][None].to(dev).contiguous()
    stream = torch.cuda.Stream(device=dev)
    st = ctypes.c_void_p(stream.cuda_stream)
    cams = _Cams(C, vm.data_ptr(), None, K.data_ptr(), 0, 4, None, None, None)
    ut = _UT(0.1, 2.0, 0.0, 0.1, 1)
    radii = torch.zeros(C, N, 2, dtype=torch.int32, device=dev)
    means2d, depths, conics = torch.zeros(C, N, 2, device=dev), torch.zeros(C, N, device=dev), torch.zeros(C, N, 3, device=dev)
    torch.cuda.synchronize()
    rc = lib.gsx_projection_ut_3dgs_fused(u32(N), _ptr(means), _ptr(quats), _ptr(scales), _ptr(opac), ctypes.byref(cams), u32(W), u32(H), f32(0.3), f32(0.01),
                                          f32(1e4), f32(0.0), ctypes.byref(ut), _ptr(radii), _ptr(means2d), _ptr(depths), _ptr(conics), None, st)
    assert rc == 0, lib.gsx_last_error()
    stream.synchronize()
    # SH colours (degree 0) on the host-computed directions, masked like rasterizer.cpp:248-262
    dirs = (means - torch.linalg.inv(vm[0].double())[:3, 3].float()).contiguous()
    mask = (radii[0] > 0).all(-1).to(torch.uint8).contiguous()
    colors = torch.zeros(N, 3, device=dev)
    torch.cuda.synchronize()
    rc = lib.gsx_spherical_harmonics_fwd(u32(0), u32(N), u32(1), _ptr(dirs), _ptr(sh), _ptr(mask), _ptr(colors), st)
    assert rc == 0, lib.gsx_last_error()
    stream.synchronize()
    colors = torch.clamp_min(colors + 0.5, 0.0).reshape(1, N, 3).contiguous()
    # binned intersection: count -> one host read -> fill
    tpg = torch.zeros(C * N, dtype=torch.int32, device=dev)
    offs = torch.zeros(C * tw * th + 1, dtype=torch.int32, device=dev)
    cws_bytes = lib.gsx_intersect_bin_count_workspace_bytes(u32(C), u32(tw), u32(th))
    cws = torch.zeros(cws_bytes + 256, dtype=torch.uint8, device=dev)
    word = torch.zeros(1, dtype=torch.int64).pin_memory()
    rc = lib.gsx_intersect_bin_count(u32(C), u32(N), _ptr(means2d), _ptr(radii), u32(16), u32(tw), u32(th), _ptr(tpg), _ptr(offs),
                                     ctypes.c_void_p(word.data_ptr()), _ptr(cws), sz(cws.numel()), st)
    assert rc == 0, lib.gsx_last_error()
    stream.synchronize()
    n_isects, max_seg = int(word.item()) & 0xFFFFFFFF, int(word.item()) >> 32
    assert n_isects == int(tpg.sum().item()) and 0 < max_seg <= n_isects
    fws_bytes = lib.gsx_intersect_bin_fill_workspace_bytes(u32(C), u32(tw), u32(th), i64(n_isects))
    fws = torch.zeros(fws_bytes + 256, dtype=torch.uint8, device=dev)
    fl = torch.zeros(n_isects, dtype=torch.int32, device=dev)
    ids = torch.zeros(n_isects, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    rc = lib.gsx_intersect_bin_fill(u32(C), u32(N), _ptr(means2d), _ptr(radii), _ptr(depths), u32(16), u32(tw), u32(th), _ptr(offs), i64(n_isects),
                                    i64(max_seg), _ptr(cws), _ptr(fl), _ptr(ids), _ptr(fws), sz(fws.numel()), st)
    assert rc == 0, lib.gsx_last_error()
    stream.synchronize()
    tpg_s, ids_s, fl_s, off_s = ops.intersect_tile_binned(means2d, radii, depths, C, 16, tw, th, True)
    assert torch.equal(tpg.reshape(C, N), tpg_s) and torch.equal(ids, ids_s) and torch.equal(fl, fl_s) and torch.equal(offs[:-1].reshape(C, th, tw), off_s)
    # too-small workspace: refused with -3, nothing launched
    rc = lib.gsx_intersect_bin_fill(u32(C), u32(N), _ptr(means2d), _ptr(radii), _ptr(depths), u32(16), u32(tw), u32(th), _ptr(offs), i64(n_isects),
                                    i64(max_seg), _ptr(cws), _ptr(fl), _ptr(ids), _ptr(fws), sz(16), st)
    assert rc == -3 and len(lib.gsx_last_error()) > 0
    # blend forward with the packed-record workspace
    opac2 = opac[None].contiguous()
    ren, alp = torch.zeros(C, H, W, 3, device=dev), torch.zeros(C, H, W, 1, device=dev)
    last = torch.zeros(C, H, W, dtype=torch.int32, device=dev)
    rws = torch.zeros(lib.gsx_rasterize_fwd_workspace_bytes(u32(C), u32(N)) + 256, dtype=torch.uint8, device=dev)
    tile_off = offs[:-1].contiguous()
    torch.cuda.synchronize()
    rc = lib.gsx_rasterize_to_pixels_from_world_3dgs_fwd(u32(N), i64(n_isects), _ptr(means), _ptr(quats), _ptr(scales), _ptr(colors), u32(3), _ptr(opac2),
                                                         _ptr(bg), None, u32(W), u32(H), u32(16), ctypes.byref(cams), ctypes.byref(ut), _ptr(tile_off),
                                                         _ptr(fl), _ptr(ren), _ptr(alp), _ptr(last), _ptr(rws), sz(rws.numel()), st)
    assert rc == 0, lib.gsx_last_error()
    stream.synchronize()
    ut_s = ops.UnscentedTransformParameters()
    common = (means, quats, scales, colors, opac2, bg, None, W, H, 16, vm, None, K, ops.CameraModelType.PINHOLE, ut_s, ops.ShutterType.GLOBAL, None, None, None,
              off_s, fl_s)
    ren_s, alp_s, last_s = ops.rasterize_to_pixels_from_world_3dgs_fwd(*common)
    assert torch.equal(ren, ren_s) and torch.equal(alp, alp_s) and torch.equal(last, last_s)
    o = oracle_pipeline(sc, frag_rel=1e-3)
    ok = o["fragile"] == 0
    assert np.abs(ren.cpu().numpy() - o["renders"])[ok].max() < 1e-4
    # blend backward with its workspace (moment records)
    g = torch.Generator().manual_seed(2)
    v_rc, v_ra = torch.randn(C, H, W, 3, generator=g).to(dev), torch.randn(C, H, W, 1, generator=g).to(dev)
    outs = [torch.full((N, 3), 7.0, device=dev), torch.full((N, 4), 7.0, device=dev), torch.full((N, 3), 7.0, device=dev), torch.full((C, N, 3), 7.0, device=dev),
            torch.full((C, N), 7.0, device=dev)]     # (pre-filled with garbage: the entry point OVERWRITES its outputs)
    bws = torch.zeros(lib.gsx_rasterize_bwd_workspace_bytes(u32(C), u32(N), i64(n_isects)) + 256, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    rc = lib.gsx_rasterize_to_pixels_from_world_3dgs_bwd(u32(N), i64(n_isects), _ptr(means), _ptr(quats), _ptr(scales), _ptr(colors), u32(3), _ptr(opac2),
                                                         _ptr(bg), None, u32(W), u32(H), u32(16), ctypes.byref(cams), ctypes.byref(ut), _ptr(tile_off),
                                                         _ptr(fl), _ptr(alp), _ptr(last), _ptr(v_rc), _ptr(v_ra), *[_ptr(t) for t in outs], _ptr(bws),
                                                         sz(bws.numel()), st)
    assert rc == 0, lib.gsx_last_error()
    stream.synchronize()
    ref = ops.rasterize_to_pixels_from_world_3dgs_bwd(*common, alp_s, last_s, v_rc, v_ra)
    for a, b in zip(outs, ref):
        # the same kernels launched twice: a Gaussian's moment records are chained in the order its tiles' workgroups retire, so its
        # sums round differently from launch to launch (tools/bwd_repeat_check.py: 1e-6 rel-L2) — entries that cancel to ~1e-6 of the
        # tensor's scale can differ by 1e-3 of themselves: compare against the tensor's scale, not element by element
        a, b = a.cpu().numpy(), b.cpu().numpy()
        assert np.abs(a - b).max() <= 2e-5 * np.abs(b).max() and np.linalg.norm(a - b) <= 1e-5 * np.linalg.norm(b)
    # unsupported request: 4 colour channels (rejected upstream too, Rasterization.cpp:65)
    rc = lib.gsx_rasterize_to_pixels_from_world_3dgs_fwd(u32(N), i64(n_isects), _ptr(means), _ptr(quats), _ptr(scales), _ptr(colors), u32(4), _ptr(opac2),
                                                         _ptr(bg), None, u32(W), u32(H), u32(16), ctypes.byref(cams), ctypes.byref(ut), _ptr(tile_off),
                                                         _ptr(fl), _ptr(ren), _ptr(alp), _ptr(last), _ptr(rws), sz(rws.numel()), st)
    assert rc < 0 and len(lib.gsx_last_error()) > 0


def _depth_ranks(lib, radii, depths, C, N):
    dev = "cuda:0"
    R, D = torch.from_numpy(radii).to(dev), torch.from_numpy(depths).to(dev)
    ranks = torch.full((C * N,), -1, dtype=torch.int32, device=dev)
    order = torch.full((C * N,), -1, dtype=torch.int32, device=dev)
    lib.gsx_intersect_depth_ranks_workspace_bytes.restype = ctypes.c_size_t
    wb = lib.gsx_intersect_depth_ranks_workspace_bytes(ctypes.c_uint32(C), ctypes.c_uint32(N))
    ws = torch.empty(wb, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    rc = lib.gsx_intersect_depth_ranks(ctypes.c_uint32(C), ctypes.c_uint32(N), _ptr(R), _ptr(D), _ptr(ranks), _ptr(order), _ptr(ws), ctypes.c_size_t(wb), None)
    assert rc == 0, lib.gsx_last_error()
    torch.cuda.synchronize()
    return ranks.cpu().numpy(), order.cpu().numpy()


@pytest.mark.parametrize("C,N,kind", [(1, 1, "random"), (1, 63, "ties"), (1, 2049, "random"), (2, 5000, "ties"), (1, 70001, "quantised"),
                                      (1, 1 << 20, "random"), (3, 349525, "ties")])
def test_depth_ranks_are_the_stable_order_of_the_depth_bits(C, N, kind, monkeypatch):
    """gsx_intersect_depth_ranks (hand-written three-pass LSD radix sort): order = stable argsort of (depth bits | culled -> 0xFFFFFFFF),
    ranks = its inverse; the library sort behind GSX_RANK_SORT=rocprim returns the same arrays."""
    lib = _lib()
    rng = np.random.default_rng(N + C)
    total = C * N
    if kind == "random":
        depths = rng.uniform(0.01, 1e3, total).astype(np.float32)
    elif kind == "ties":
        depths = rng.integers(1, 40, total).astype(np.float32) * 0.25          # long runs of equal keys: stability decides
    else:
        depths = (np.exp(rng.uniform(-4, 6, total)) * 64).round().astype(np.float32) / 64
    radii = rng.integers(0, 3, (total, 2)).astype(np.int32)                    # 5/9 of the Gaussians culled
    keys = np.where((radii > 0).all(-1), depths.view(np.uint32), np.uint32(0xFFFFFFFF))
    want_order = np.argsort(keys, kind="stable").astype(np.int32)
    want_ranks = np.empty(total, np.int32)
    want_ranks[want_order] = np.arange(total, dtype=np.int32)
    ranks, order = _depth_ranks(lib, radii, depths, C, N)
    np.testing.assert_array_equal(order, want_order)
    np.testing.assert_array_equal(ranks, want_ranks)
    monkeypatch.setenv("GSX_RANK_SORT", "rocprim")
    ranks2, order2 = _depth_ranks(lib, radii, depths, C, N)
    np.testing.assert_array_equal(order2, want_order)
    np.testing.assert_array_equal(ranks2, want_ranks)
