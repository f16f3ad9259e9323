"""ctypes loader for the CPU oracle (oracle/gsx_oracle.cpp).

TEST INFRASTRUCTURE ONLY.  May be imported from tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py — never from the product package.

All functions take / return numpy arrays; dtype float32 selects the `_f32` entry points (reference
operation order in fp32), float64 the `_f64` ones (same formulas re-evaluated in double).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgsx_oracle.so")
_lib = None

PINHOLE, ORTHO, FISHEYE = 0, 1, 2
SHUTTER_GLOBAL = 4


def build(force=False):
    src = os.path.join(_HERE, "gsx_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libgsx_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.gsx_oracle_isect_count_f32.restype = ctypes.c_int64
        _lib.gsx_oracle_isect_count_f64.restype = ctypes.c_int64
    return _lib


def _suf(dtype):
    return "f32" if np.dtype(dtype) == np.float32 else "f64"


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype):
    return None if a is None else np.ascontiguousarray(a, dtype=dtype)


def _fl(dtype, v):
    return ctypes.c_float(v) if np.dtype(dtype) == np.float32 else ctypes.c_double(v)


def quat_to_rotmat(quats):
    dt = quats.dtype
    q = _c(quats, dt).reshape(-1, 4)
    out = np.empty((q.shape[0], 3, 3), dt)
    getattr(lib(), "gsx_oracle_quat_to_rotmat_" + _suf(dt))(ctypes.c_int64(q.shape[0]), _p(q), _p(out))
    return out


def preci_half(quats, scales):
    """M = diag(1/s) R^T of the blend kernels (quat_scale_to_preci_half): M^T M is the reference's precision matrix."""
    dt = quats.dtype
    q, sc = _c(quats, dt).reshape(-1, 4), _c(scales, dt).reshape(-1, 3)
    out = np.empty((q.shape[0], 3, 3), dt)
    getattr(lib(), "gsx_oracle_preci_half_" + _suf(dt))(ctypes.c_int64(q.shape[0]), _p(q), _p(sc), _p(out))
    return out


def projection_ut(means, quats, scales, opacities, viewmats0, Ks, width, height, eps2d=0.3, near_plane=0.01,
                  far_plane=1e4, radius_clip=0.0, calc_compensations=False, camera_model=PINHOLE,
                  ut=(0.1, 2.0, 0.0, 0.1, True), shutter=SHUTTER_GLOBAL, viewmats1=None, radial=None,
                  tangential=None, thin_prism=None):
    dt = means.dtype
    means, quats, scales = _c(means, dt), _c(quats, dt), _c(scales, dt)
    opacities = _c(opacities, dt)
    viewmats0, Ks, viewmats1 = _c(viewmats0, dt), _c(Ks, dt), _c(viewmats1, dt)
    radial, tangential, thin_prism = _c(radial, dt), _c(tangential, dt), _c(thin_prism, dt)
    N, C = means.shape[0], Ks.shape[0]
    radii = np.zeros((C, N, 2), np.int32)
    means2d = np.zeros((C, N, 2), dt)
    depths = np.zeros((C, N), dt)
    conics = np.zeros((C, N, 3), dt)
    comp = np.zeros((C, N), dt) if calc_compensations else None
    getattr(lib(), "gsx_oracle_projection_ut_" + _suf(dt))(
        ctypes.c_uint32(C), ctypes.c_uint32(N), _p(means), _p(quats), _p(scales), _p(opacities), _p(viewmats0),
        _p(viewmats1), _p(Ks), ctypes.c_uint32(width), ctypes.c_uint32(height), _fl(dt, eps2d), _fl(dt, near_plane),
        _fl(dt, far_plane), _fl(dt, radius_clip), ctypes.c_int(camera_model), _fl(dt, ut[0]), _fl(dt, ut[1]),
        _fl(dt, ut[2]), _fl(dt, ut[3]), ctypes.c_int(int(ut[4])), ctypes.c_int(shutter), _p(radial), _p(tangential),
        _p(thin_prism), _p(radii), _p(means2d), _p(depths), _p(conics), _p(comp))
    return radii, means2d, depths, conics, comp


def sh_fwd(degree, dirs, coeffs, masks=None):
    dt = dirs.dtype
    dirs, coeffs = _c(dirs, dt), _c(coeffs, dt)
    K = coeffs.shape[-2]
    N = dirs.size // 3
    masks = None if masks is None else np.ascontiguousarray(masks, dtype=np.uint8)
    colors = np.zeros(dirs.shape, dt)
    getattr(lib(), "gsx_oracle_sh_fwd_" + _suf(dt))(ctypes.c_uint32(N), ctypes.c_uint32(K), ctypes.c_uint32(degree),
                                                    _p(dirs), _p(coeffs), _p(masks), _p(colors))
    return colors


def sh_bwd(degree, dirs, coeffs, masks, v_colors, compute_v_dirs=True):
    dt = dirs.dtype
    dirs, coeffs, v_colors = _c(dirs, dt), _c(coeffs, dt), _c(v_colors, dt)
    K = coeffs.shape[-2]
    N = dirs.size // 3
    masks = None if masks is None else np.ascontiguousarray(masks, dtype=np.uint8)
    v_coeffs = np.zeros(coeffs.shape, dt)
    v_dirs = np.zeros(dirs.shape, dt) if compute_v_dirs else None
    getattr(lib(), "gsx_oracle_sh_bwd_" + _suf(dt))(ctypes.c_uint32(N), ctypes.c_uint32(K), ctypes.c_uint32(degree),
                                                    _p(dirs), _p(coeffs), _p(masks), _p(v_colors), _p(v_coeffs),
                                                    _p(v_dirs))
    return v_coeffs, v_dirs


def intersect_tile(means2d, radii, depths, C, tile_size, tile_width, tile_height, sort=True):
    dt = means2d.dtype
    means2d, depths = _c(means2d, dt), _c(depths, dt)
    radii = _c(radii, np.int32)
    N = means2d.size // 2 // C
    tpg = np.zeros((C, N), np.int32)
    suf = _suf(dt)
    n = getattr(lib(), "gsx_oracle_isect_count_" + suf)(ctypes.c_uint32(C), ctypes.c_uint32(N), _p(means2d),
                                                        _p(radii), ctypes.c_uint32(tile_size),
                                                        ctypes.c_uint32(tile_width), ctypes.c_uint32(tile_height),
                                                        _p(tpg))
    isect_ids = np.zeros((n,), np.int64)
    flatten_ids = np.zeros((n,), np.int32)
    if n:
        getattr(lib(), "gsx_oracle_isect_fill_" + suf)(ctypes.c_uint32(C), ctypes.c_uint32(N), _p(means2d), _p(radii),
                                                       _p(depths), ctypes.c_uint32(tile_size),
                                                       ctypes.c_uint32(tile_width), ctypes.c_uint32(tile_height),
                                                       ctypes.c_int(int(sort)), _p(isect_ids), _p(flatten_ids))
    return tpg, isect_ids, flatten_ids


def intersect_offset(isect_ids, C, tile_width, tile_height):
    isect_ids = _c(isect_ids, np.int64)
    offsets = np.zeros((C, tile_height, tile_width), np.int32)
    lib().gsx_oracle_isect_offsets(ctypes.c_int64(isect_ids.shape[0]), _p(isect_ids), ctypes.c_uint32(C),
                                   ctypes.c_uint32(tile_width), ctypes.c_uint32(tile_height), _p(offsets))
    return offsets


def _cam_args(dt, viewmats0, viewmats1, Ks, camera_model, shutter, radial, tangential, thin_prism):
    return (_c(viewmats0, dt), _c(viewmats1, dt), _c(Ks, dt), camera_model, shutter, _c(radial, dt),
            _c(tangential, dt), _c(thin_prism, dt))


def rasterize_fwd(means, quats, scales, colors, opacities, backgrounds, masks, width, height, tile_size, viewmats0, Ks,
                  tile_offsets, flatten_ids, camera_model=PINHOLE, shutter=SHUTTER_GLOBAL, viewmats1=None, radial=None,
                  tangential=None, thin_prism=None, frag_rel=None):
    dt = means.dtype
    means, quats, scales, colors, opacities = (_c(a, dt) for a in (means, quats, scales, colors, opacities))
    backgrounds = None if backgrounds is None or backgrounds.size == 0 else _c(backgrounds, dt)
    masks = None if masks is None else np.ascontiguousarray(masks, dtype=np.uint8)
    vm0, vm1, Ks, cm, sh, rad, tan, pri = _cam_args(dt, viewmats0, viewmats1, Ks, camera_model, shutter, radial,
                                                    tangential, thin_prism)
    tile_offsets, flatten_ids = _c(tile_offsets, np.int32), _c(flatten_ids, np.int32)
    C, N = tile_offsets.shape[0], means.shape[0]
    renders = np.zeros((C, height, width, 3), dt)
    alphas = np.zeros((C, height, width, 1), dt)
    last_ids = np.zeros((C, height, width), np.int32)
    fragile = np.zeros((C, height, width), np.uint8) if frag_rel is not None else None
    getattr(lib(), "gsx_oracle_raster_fwd_" + _suf(dt))(
        ctypes.c_uint32(C), ctypes.c_uint32(N), ctypes.c_int64(flatten_ids.shape[0]), _p(means), _p(quats), _p(scales),
        _p(colors), _p(opacities), _p(backgrounds), _p(masks), ctypes.c_uint32(width), ctypes.c_uint32(height),
        ctypes.c_uint32(tile_size), _p(vm0), _p(vm1), _p(Ks), ctypes.c_int(cm), ctypes.c_int(sh), _p(rad), _p(tan),
        _p(pri), _p(tile_offsets), _p(flatten_ids), _p(renders), _p(alphas), _p(last_ids), _p(fragile),
        _fl(dt, frag_rel if frag_rel is not None else 0.0))
    if frag_rel is not None:
        return renders, alphas, last_ids, fragile
    return renders, alphas, last_ids


def rasterize_bwd(means, quats, scales, colors, opacities, backgrounds, masks, width, height, tile_size, viewmats0, Ks,
                  tile_offsets, flatten_ids, render_alphas, last_ids, v_render_colors, v_render_alphas,
                  camera_model=PINHOLE, shutter=SHUTTER_GLOBAL, viewmats1=None, radial=None, tangential=None,
                  thin_prism=None):
    dt = means.dtype
    means, quats, scales, colors, opacities = (_c(a, dt) for a in (means, quats, scales, colors, opacities))
    backgrounds = None if backgrounds is None or backgrounds.size == 0 else _c(backgrounds, dt)
    masks = None if masks is None else np.ascontiguousarray(masks, dtype=np.uint8)
    vm0, vm1, Ks, cm, sh, rad, tan, pri = _cam_args(dt, viewmats0, viewmats1, Ks, camera_model, shutter, radial,
                                                    tangential, thin_prism)
    tile_offsets, flatten_ids = _c(tile_offsets, np.int32), _c(flatten_ids, np.int32)
    render_alphas, v_render_colors, v_render_alphas = (_c(a, dt) for a in (render_alphas, v_render_colors,
                                                                            v_render_alphas))
    last_ids = _c(last_ids, np.int32)
    C, N = tile_offsets.shape[0], means.shape[0]
    v_means, v_quats, v_scales = np.zeros((N, 3), dt), np.zeros((N, 4), dt), np.zeros((N, 3), dt)
    v_colors, v_opac = np.zeros((C, N, 3), dt), np.zeros((C, N), dt)
    getattr(lib(), "gsx_oracle_raster_bwd_" + _suf(dt))(
        ctypes.c_uint32(C), ctypes.c_uint32(N), ctypes.c_int64(flatten_ids.shape[0]), _p(means), _p(quats), _p(scales),
        _p(colors), _p(opacities), _p(backgrounds), _p(masks), ctypes.c_uint32(width), ctypes.c_uint32(height),
        ctypes.c_uint32(tile_size), _p(vm0), _p(vm1), _p(Ks), ctypes.c_int(cm), ctypes.c_int(sh), _p(rad), _p(tan),
        _p(pri), _p(tile_offsets), _p(flatten_ids), _p(render_alphas), _p(last_ids), _p(v_render_colors),
        _p(v_render_alphas), _p(v_means), _p(v_quats), _p(v_scales), _p(v_colors), _p(v_opac))
    return v_means, v_quats, v_scales, v_colors, v_opac


def relocation(opacities, scales, ratios, binoms, n_max):
    dt = opacities.dtype
    o, s, b = _c(opacities, dt), _c(scales, dt), _c(binoms, dt)
    r = _c(ratios, np.int32)
    new_o, new_s = np.zeros_like(o), np.zeros_like(s)
    getattr(lib(), "gsx_oracle_relocation_" + _suf(dt))(ctypes.c_int64(o.shape[0]), _p(o), _p(s), _p(r), _p(b), ctypes.c_int(n_max),
                                                        _p(new_o), _p(new_s))
    return new_o, new_s


def add_noise(raw_opacities, raw_scales, raw_quats, noise, means, lr):
    dt = means.dtype
    out = np.ascontiguousarray(means, dtype=dt).copy()
    getattr(lib(), "gsx_oracle_add_noise_" + _suf(dt))(ctypes.c_int64(out.shape[0]), _p(_c(raw_opacities, dt)), _p(_c(raw_scales, dt)),
                                                       _p(_c(raw_quats, dt)), _p(_c(noise, dt)), _p(out), _fl(dt, lr))
    return out


def adam_step(param, exp_avg, exp_avg_sq, grad, lr, beta1, beta2, eps, step_count):
    """One reference Adam step (in place on copies; returns param, exp_avg, exp_avg_sq)."""
    dt = param.dtype
    p, m, v = (np.ascontiguousarray(a, dtype=dt).copy() for a in (param, exp_avg, exp_avg_sq))
    bc1 = 1.0 / (1.0 - beta1 ** step_count)
    bc2 = 1.0 / np.sqrt(1.0 - beta2 ** step_count)
    getattr(lib(), "gsx_oracle_adam_step_" + _suf(dt))(ctypes.c_int64(p.size), _p(p), _p(m), _p(v), _p(_c(grad, dt)), _fl(dt, lr),
                                                       _fl(dt, beta1), _fl(dt, beta2), _fl(dt, eps), _fl(dt, bc1), _fl(dt, bc2))
    return p, m, v


SSIM_C1, SSIM_C2 = 0.01 * 0.01, 0.03 * 0.03


def fused_ssim_fwd(img1, img2, train=True):
    """fusedssim (ssim.cu:436-470) on [B,CH,H,W]: (map, dm_dmu1, dm_dsigma1_sq, dm_dsigma12)."""
    dt = img1.dtype
    a, b = _c(img1, dt), _c(img2, dt)
    B, CH, H, W = a.shape
    outs = [np.zeros_like(a) for _ in range(4 if train else 1)]
    getattr(lib(), "gsx_oracle_fused_ssim_fwd_" + _suf(dt))(
        ctypes.c_int64(B), ctypes.c_int64(CH), ctypes.c_int64(H), ctypes.c_int64(W), _fl(dt, SSIM_C1), _fl(dt, SSIM_C2), _p(a), _p(b),
        _p(outs[0]), *( [_p(o) for o in outs[1:]] if train else [None, None, None]))
    return tuple(outs) if train else (outs[0], None, None, None)


def fused_ssim_bwd(img1, img2, dL_dmap, dm1, ds1, ds12):
    dt = img1.dtype
    a, b = _c(img1, dt), _c(img2, dt)
    B, CH, H, W = a.shape
    out = np.zeros_like(a)
    getattr(lib(), "gsx_oracle_fused_ssim_bwd_" + _suf(dt))(
        ctypes.c_int64(B), ctypes.c_int64(CH), ctypes.c_int64(H), ctypes.c_int64(W), _p(a), _p(b), _p(_c(dL_dmap, dt)), _p(out),
        _p(_c(dm1, dt)), _p(_c(ds1, dt)), _p(_c(ds12, dt)))
    return out


def photometric_loss(render_hwc, gt_chw, lambda_dssim=0.2):
    """trainer.cpp:103-127 on the clamped image (rasterizer.cpp:401): returns (loss, l1, ssim_mean, d loss / d render_hwc).
    render_hwc [C,H,W,3] unclamped blend output, gt_chw [C,3,H,W]."""
    dt = render_hwc.dtype
    raw = _c(render_hwc, dt)
    img = np.clip(raw, 0, 1).transpose(0, 3, 1, 2).copy()
    gt = _c(gt_chw, dt)
    C, _, H, W = img.shape
    m, dm1, ds1, ds12 = fused_ssim_fwd(img, gt, True)
    crop = 5 if (H > 10 and W > 10) else 0
    valid = m[:, :, crop:H - crop, crop:W - crop]
    l1 = np.abs(img - gt).mean(dtype=np.float64)
    ssim = valid.mean(dtype=np.float64)
    loss = (1 - lambda_dssim) * l1 + lambda_dssim * (1 - ssim)
    dmap = np.zeros_like(img)
    if crop:  # fused_ssim.cuh:85-96: without a crop the scattered dL/dmap stays zero (upstream quirk, kept)
        dmap[:, :, crop:H - crop, crop:W - crop] = -lambda_dssim / valid.size
    g = fused_ssim_bwd(img, gt, dmap, dm1, ds1, ds12) + (1 - lambda_dssim) / img.size * np.sign(img - gt)
    g = g.transpose(0, 2, 3, 1) * ((raw >= 0) & (raw <= 1))
    return float(loss), float(l1), float(ssim), np.ascontiguousarray(g, dtype=dt)
