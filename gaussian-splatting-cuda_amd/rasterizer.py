"""Render glue + autograd wrappers around the gsplat operators: the Python mirror of the reference's
`gs::training::rasterize` (src/training/rasterization/rasterizer.cpp:46-437) and of the three wrappers in
src/training/rasterization/rasterizer_autograd.cpp (SphericalHarmonicsFunction :12-132,
fully_fused_projection_with_ut :135-265, GUTRasterizationFunction :267-391).

Data contracts mirrored from the reference:
  * Camera: row-major world->camera `viewmat` [1,4,4] (src/core/camera.cpp:15-22), K [1,3,3] (:80-89)
  * SplatData activations (src/core/splat_data.cpp:267-286): opacity = sigmoid(raw).squeeze(-1),
    rotation = normalize(raw), scaling = exp(raw), shs = cat(sh0, shN)
Hot-path constants hard-coded upstream (rasterizer.cpp:176-181): eps2d 0.3, near 0.01, far 1e4,
radius_clip 0, tile 16, calc_compensations = antialiased, GLOBAL shutter, default UT parameters.
"""
from dataclasses import dataclass, field
import os
import threading
from typing import Optional

import torch

from . import ops

TILE_SIZE = 16
EPS2D, NEAR_PLANE, FAR_PLANE, RADIUS_CLIP = 0.3, 0.01, 10000.0, 0.0
# rasterize_fused: one kernel for activations -> projection -> SH colours -> packed blend records (ops.frontend_fused); False = the
# separate launches (A/B tools, tests)
FUSED_FRONTEND = os.environ.get("GSX_FUSED_FRONTEND", "1") != "0"
# rasterize_fused backward: the activation Jacobians as the epilogue of the blend backward's gather kernel (ops.rasterize_bwd_act); False = the
# separate splat_activations_bwd launch (A/B tools, tests — same values)
ACT_EPILOGUE = os.environ.get("GSX_ACT_EPILOGUE", "1") != "0"

# rasterize_fused: size of the tiles the intersection LISTS are built for.  16 = the reference's; 32 = one list per 2 x 2 pixel tiles
# (include/gsx.h, the blend entry points' `tile_size`): same image, the intersection handles ~3x fewer keys when the Gaussians cover
# many tiles (trained dense scenes), at the price of longer candidate lists per pixel tile when they are small (S-1M).  Chosen per
# problem shape from the previous frame's list density, with hysteresis; GSX_LIST_TILE=16|32 forces one (tests, A/B tools).
_LIST_TILE_STATE = {}
# keys per 16-px tile above which 32-px lists pay / keys per 32-px tile below which they stop paying.  The same frame has ~1.3x the keys
# per 32-px tile that it has per 16-px tile (3.1x fewer keys on 4x fewer tiles), so UP = 3000 enters at ~3900 keys per 32-px tile and
# DOWN = 2500 leaves at ~1900 keys per 16-px tile: a band of 1.5x, no frame-to-frame flipping
LIST_TILE_UP, LIST_TILE_DOWN = 3000.0, 2500.0


class IsectCapacityMiss(RuntimeError):
    """Guarded lists (rasterize_fused(guarded=True)): this frame's intersections outgrew the capacity the optimistic fill was launched
    with, so the kernels rendered it with EMPTY lists.  Raised from the render's backward BEFORE anything irreversible ran (the SH
    tensor's fused Adam step, a gradient exchange); the capacity hint has been raised: run the iteration again (trainer.Trainer and
    bench.py do)."""


def _policy_key(width, height, device_index, n_gaussians):
    """Key of the list-granularity / record-layout statistics: image shape, device and a COARSE size class of the model (its bit length:
    a model growing by 5 % every hundred iterations keeps its key — it changes at doublings only — while two models of very different
    size rendered at one resolution in the same process, e.g. training and the evaluation of another model, no longer steer each
    other's choices: ADVICE r04)."""
    return (width, height, device_index, int(n_gaussians).bit_length())


def _list_tile_for(key):
    forced = os.environ.get("GSX_LIST_TILE")
    if forced in ("16", "32"):
        return int(forced)
    return _LIST_TILE_STATE.get(key, 16)


# list entries per Gaussian of the previous frame of a problem shape (at its list granularity): above RANGES_ABOVE the next frame keeps the
# backward's records in per-Gaussian runs even on 16-pixel lists (the same threshold at which the chained layout goes to four chains)
_ENTRIES_PER_GAUSSIAN = {}
RANGES_ABOVE = 8.0


def _record_ranges_for(key):
    return _list_tile_for(key) != TILE_SIZE or _ENTRIES_PER_GAUSSIAN.get(key, 0.0) > RANGES_ABOVE


def _list_tile_update(key, list_tile, n_isects, n_list_tiles, n_gaussians=0):
    if n_gaussians > 0:
        _ENTRIES_PER_GAUSSIAN[key] = n_isects / n_gaussians
    density = n_isects / max(1, n_list_tiles)
    if list_tile == 16 and density >= LIST_TILE_UP:
        _LIST_TILE_STATE[key] = 32
    elif list_tile == 32 and density < LIST_TILE_DOWN:
        _LIST_TILE_STATE[key] = 16


@dataclass
class Camera:
    viewmat: torch.Tensor                  # [4,4] row-major world->camera
    K: torch.Tensor                        # [3,3]
    width: int
    height: int
    camera_model: object = None            # ops.CameraModelType, default PINHOLE
    radial: Optional[torch.Tensor] = None  # 1-D, padded to >=4 like rasterizer.cpp:183-195
    tangential: Optional[torch.Tensor] = None

    def world_view_transform(self):
        return self.viewmat.reshape(1, 4, 4)

    def K_batched(self):
        return self.K.reshape(1, 3, 3)


@dataclass
class SplatData:
    """Raw (pre-activation) parameters.  The reference keeps SH as two tensors `sh0 [N,1,3]` / `shN [N,K-1,3]` and
    concatenates them every frame (`get_shs`, splat_data.cpp:284-286: a 192 MB copy per frame at 1M/deg 3, plus the
    split in backward).  Here they are views of ONE `sh [N,K,3]` leaf, so `get_shs()` is free; a per-group learning
    rate can still address `sh[:, :1]` and `sh[:, 1:]`."""
    means: torch.Tensor         # [N,3]
    sh: torch.Tensor            # [N,K,3]  (sh0 = sh[:, :1], shN = sh[:, 1:])
    scaling_raw: torch.Tensor   # [N,3] log-scales
    rotation_raw: torch.Tensor  # [N,4] wxyz
    opacity_raw: torch.Tensor   # [N,1] logits
    active_sh_degree: int = 3

    def params(self):
        return [self.means, self.sh, self.scaling_raw, self.rotation_raw, self.opacity_raw]

    @property
    def sh0(self): return self.sh[:, :1]
    @property
    def shN(self): return self.sh[:, 1:]

    def get_means(self): return self.means
    def get_opacity(self): return torch.sigmoid(self.opacity_raw).squeeze(-1)
    def get_rotation(self): return torch.nn.functional.normalize(self.rotation_raw, dim=-1)
    def get_scaling(self): return torch.exp(self.scaling_raw)
    def get_shs(self): return self.sh


@dataclass
class RenderOutput:
    _image: torch.Tensor = None
    render_hwc: torch.Tensor = None
    alpha: torch.Tensor = None
    depth: torch.Tensor = None
    means2d: torch.Tensor = None
    depths: torch.Tensor = None
    radii: torch.Tensor = None
    visibility: torch.Tensor = None
    width: int = 0
    height: int = 0
    _n_isects: int = 0
    lists: object = None   # ops.IsectLists handle of a guarded render (rasterize_fused(guarded=True)), else None
    aux: dict = field(default_factory=dict)

    @property
    def n_isects(self):
        """Number of tile intersections of the frame; on a guarded render this reads the count the GPU copied to the host (waits for
        it if the GPU has not passed the intersection yet)."""
        return self._n_isects if self.lists is None else int(self.lists.confirm()[0])

    @n_isects.setter
    def n_isects(self, value):
        self._n_isects = int(value)

    def confirm(self):
        """Guarded render: True when the frame's lists were complete (the image is the frame's image); False = the capacity was
        exceeded and the kernels rendered EMPTY lists: render again (the hint has been raised).  Always True on an exact render."""
        return True if self.lists is None else bool(self.lists.confirm()[2])

    @property
    def image(self):
        """[3,H,W] clamped to [0,1] (rasterizer.cpp:401); computed lazily from the blend's [1,H,W,3] output when only that was kept."""
        if self._image is None and self.render_hwc is not None:
            self._image = torch.clamp(self.render_hwc.squeeze(0).permute(2, 0, 1), 0.0, 1.0)
        return self._image

    @image.setter
    def image(self, value):
        self._image = value


class SphericalHarmonicsFunction(torch.autograd.Function):
    """rasterizer_autograd.cpp:12-132 (the degree is passed as a plain int: upstream round-trips it through a
    1-element device tensor and .item()s it back, a per-frame H2D+D2H with no effect on results)."""

    @staticmethod
    def forward(ctx, sh_degree, dirs, coeffs, masks):
        num = (sh_degree + 1) ** 2
        assert dirs.shape[-1] == 3 and coeffs.shape[-1] == 3 and coeffs.shape[-2] >= num
        assert dirs.shape[:-1] == coeffs.shape[:-2], "dirs and coeffs batch dimensions must match"
        dirs = dirs.contiguous()
        coeffs = coeffs.contiguous()
        if masks is None:
            masks = torch.ones(dirs.shape[:-1], dtype=torch.bool, device=dirs.device)
        masks = masks.contiguous()
        colors = ops.spherical_harmonics_fwd(sh_degree, dirs.reshape(-1, 3), coeffs.reshape(-1, coeffs.shape[-2], 3),
                                             masks.reshape(-1))
        ctx.save_for_backward(dirs, coeffs, masks)
        ctx.sh_degree = sh_degree
        return colors.reshape(dirs.shape)

    @staticmethod
    def backward(ctx, v_colors):
        dirs, coeffs, masks = ctx.saved_tensors
        K = coeffs.shape[-2]
        compute_v_dirs = ctx.needs_input_grad[1]
        v_coeffs, v_dirs = ops.spherical_harmonics_bwd(K, ctx.sh_degree, dirs.reshape(-1, 3), coeffs.reshape(-1, K, 3),
                                                       masks.reshape(-1), v_colors.contiguous().reshape(-1, 3),
                                                       compute_v_dirs)
        v_dirs = v_dirs.reshape(dirs.shape) if (compute_v_dirs and v_dirs is not None) else None
        v_coeffs = v_coeffs.reshape(coeffs.shape) if ctx.needs_input_grad[2] else None
        return None, v_dirs, v_coeffs, None


def spherical_harmonics(sh_degree, dirs, coeffs, masks=None):
    if coeffs.shape[:-2] != dirs.shape[:-1]:  # broadcast [1,N,K,3] -> [C,N,K,3] (rasterizer.cpp:260)
        coeffs = coeffs.expand(*dirs.shape[:-1], *coeffs.shape[-2:])
    return SphericalHarmonicsFunction.apply(sh_degree, dirs, coeffs, masks)


def fully_fused_projection_with_ut(means, quats, scales, opacities, viewmat, K, radial, tangential, thin_prism,
                                   width, height, scaling_modifier=1.0, camera_model=None, ut_params=None,
                                   eps2d=EPS2D, near_plane=NEAR_PLANE, far_plane=FAR_PLANE, radius_clip=RADIUS_CLIP):
    """rasterizer_autograd.cpp:135-265 — non-differentiable (no autograd node upstream either)."""
    camera_model = camera_model if camera_model is not None else ops.CameraModelType.PINHOLE
    ut_params = ut_params or ops.UnscentedTransformParameters()
    with torch.no_grad():
        scaled = (scales * scaling_modifier).contiguous()
        return ops.projection_ut_3dgs_fused(means.contiguous(), quats.contiguous(), scaled,
                                            opacities.contiguous() if opacities is not None else None,
                                            viewmat.contiguous(), None, K.contiguous(), width, height, eps2d, near_plane,
                                            far_plane, radius_clip, False, camera_model, ut_params,
                                            ops.ShutterType.GLOBAL, radial, tangential, thin_prism)


class GUTRasterizationFunction(torch.autograd.Function):
    """rasterizer_autograd.cpp:267-391."""

    @staticmethod
    def forward(ctx, means, quats, scales, colors, opacities, bg_color, masks, viewmat, K, radial, tangential,
                thin_prism, isect_offsets, flatten_ids, width, height, tile_size, scaling_modifier, camera_model,
                ut_params):
        assert colors.shape[-1] == 3, "only RGB (3 channels) is supported on this path (rasterizer_autograd.cpp:285)"
        scales = (scales * scaling_modifier).contiguous()
        means, quats, colors, opacities = means.contiguous(), quats.contiguous(), colors.contiguous(), opacities.contiguous()
        bg = bg_color.contiguous() if (bg_color is not None and bg_color.numel() > 0) else None
        renders, alphas, last_ids = ops.rasterize_to_pixels_from_world_3dgs_fwd(
            means, quats, scales, colors, opacities, bg, masks, width, height, tile_size, viewmat.contiguous(), None,
            K.contiguous(), camera_model, ut_params, ops.ShutterType.GLOBAL, radial, tangential, thin_prism,
            isect_offsets.contiguous(), flatten_ids.contiguous())
        ctx.save_for_backward(means, quats, scales, colors, opacities, viewmat, K, isect_offsets, flatten_ids, alphas,
                              last_ids)
        ctx.extra = (bg, masks, radial, tangential, thin_prism, width, height, tile_size, camera_model, ut_params,
                     scaling_modifier)
        return renders, alphas

    @staticmethod
    def backward(ctx, v_render_colors, v_render_alpha):
        means, quats, scales, colors, opacities, viewmat, K, isect_offsets, flatten_ids, alphas, last_ids = ctx.saved_tensors
        bg, masks, radial, tangential, thin_prism, width, height, tile_size, camera_model, ut_params, _ = ctx.extra
        v_render_colors = v_render_colors.contiguous()
        v_render_alpha = v_render_alpha.contiguous()
        v_means, v_quats, v_scales, v_colors, v_opac = ops.rasterize_to_pixels_from_world_3dgs_bwd(
            means, quats, scales, colors, opacities, bg, masks, width, height, tile_size, viewmat.contiguous(), None,
            K.contiguous(), camera_model, ut_params, ops.ShutterType.GLOBAL, radial, tangential, thin_prism,
            isect_offsets.contiguous(), flatten_ids.contiguous(), alphas, last_ids, v_render_colors, v_render_alpha)
        v_bg = None
        if bg is not None and ctx.needs_input_grad[5]:
            v_bg = (v_render_colors * (1.0 - alphas)).float().sum(dim=(-3, -2))
        # NB: v_scales is w.r.t. the scaled scales and is returned as is (upstream does not multiply by
        # scaling_modifier either: rasterizer_autograd.cpp:291,377).
        return (v_means, v_quats, v_scales, v_colors, v_opac, v_bg) + (None,) * 14


def _distortion_args(camera: Camera, cam_model, device):
    """Distortion coefficients as the kernels read them (rasterizer.cpp:183-195): radial padded to >= 4 (6 for the OpenCV
    pinhole model, Cameras.cuh:483), tangential to 2; None when the camera has none."""
    def _pad(t, n):
        if t is None or t.numel() == 0:
            return None
        t = t.reshape(-1).to(device, torch.float32)
        if t.numel() < n:
            t = torch.nn.functional.pad(t, (0, n - t.numel()))
        return t.contiguous()
    radial, tangential = _pad(camera.radial, 4), _pad(camera.tangential, 2)
    if radial is not None and cam_model == ops.CameraModelType.PINHOLE and radial.numel() < 6:
        radial = torch.nn.functional.pad(radial, (0, 6 - radial.numel()))
    return radial, tangential


def rasterize(camera: Camera, model: SplatData, bg_color: Optional[torch.Tensor], scaling_modifier: float = 1.0,
              packed: bool = False, antialiased: bool = False, sh_degree: Optional[int] = None) -> RenderOutput:
    """gs::training::rasterize, RGB render mode (the only one that works on this path upstream)."""
    assert not packed, "Packed mode is not supported in this implementation"
    W, H = int(camera.width), int(camera.height)
    viewmat = camera.world_view_transform()
    K = camera.K_batched()
    means = model.get_means()
    opacities = model.get_opacity()
    scales = model.get_scaling()
    rotations = model.get_rotation()
    sh_coeffs = model.get_shs()
    sh_degree = model.active_sh_degree if sh_degree is None else sh_degree
    assert sh_coeffs.shape[1] >= (sh_degree + 1) ** 2, "Not enough SH coefficients"
    cam_model = camera.camera_model if camera.camera_model is not None else ops.CameraModelType.PINHOLE
    ut = ops.UnscentedTransformParameters()

    radial, tangential = _distortion_args(camera, cam_model, means.device)

    # 1. projection (no grad)
    radii, means2d, depths, conics, _ = fully_fused_projection_with_ut(
        means, rotations, scales, opacities, viewmat, K, radial, tangential, None, W, H, scaling_modifier, cam_model, ut)
    # 2. colours from SH
    campos = torch.inverse(viewmat)[:, :3, 3]
    dirs = means.unsqueeze(0) - campos.unsqueeze(1)
    masks = (radii > 0).all(-1)
    colors = spherical_harmonics(sh_degree, dirs, sh_coeffs.unsqueeze(0), masks)
    colors = torch.clamp_min(colors + 0.5, 0.0)
    # 3. background / opacities
    bg = bg_color.reshape(1, -1).to(means.device) if (bg_color is not None and bg_color.numel() > 0) else None
    final_opacities = opacities.unsqueeze(0)
    # 4. tile intersection
    tw, th = (W + TILE_SIZE - 1) // TILE_SIZE, (H + TILE_SIZE - 1) // TILE_SIZE
    with torch.no_grad():
        _, isect_ids, flatten_ids = ops.intersect_tile(means2d, radii, depths, None, None, 1, TILE_SIZE, tw, th, True)
        isect_offsets = ops.intersect_offset(isect_ids, 1, tw, th).reshape(1, th, tw)
    # 5. blend
    renders, alphas = GUTRasterizationFunction.apply(means, rotations, scales, colors, final_opacities, bg, None, viewmat, K,
                                                     radial, tangential, None, isect_offsets, flatten_ids, W, H, TILE_SIZE,
                                                     scaling_modifier, cam_model, ut)
    out = RenderOutput()
    out.image = torch.clamp(renders.squeeze(0).permute(2, 0, 1), 0.0, 1.0)
    out.alpha = alphas.squeeze(0).permute(2, 0, 1)
    out.means2d = means2d
    out.depths = depths.squeeze(0)
    out.radii = radii.squeeze(0).max(-1).values
    out.visibility = out.radii > 0
    out.width, out.height = W, H
    out.n_isects = int(flatten_ids.shape[0])
    out.aux = dict(isect_offsets=isect_offsets, flatten_ids=flatten_ids, colors=colors, radii_full=radii)
    return out


# ---------------------------------------------------------------------------------------------------------
# Fused render: the same pipeline as rasterize(), RGB mode, with the chains of small torch ops replaced by the
# fused glue kernels (activations, campos/dirs/masks/SH/+0.5/clamp) and ONE autograd node for the whole render.
# Gradients can be written straight into caller-provided buffers (`grad_sinks`, e.g. the views of a flat
# all-reduce bucket): no zero fill and no AccumulateGrad add pass over the 192 MB SH gradient.
# ---------------------------------------------------------------------------------------------------------
_TLS = threading.local()   # the guarded-lists handle of the render this THREAD just ran: forward() -> rasterize_fused (a handle is not a tensor output)


class GutRenderFunction(torch.autograd.Function):

    @staticmethod
    def forward(ctx, means, sh, scaling_raw, rotation_raw, opacity_raw, viewmat, K, bg, width, height, sh_degree, scaling_modifier,
                camera_model, radial, tangential, grad_sinks, guarded=False):
        ut = ops.UnscentedTransformParameters()
        means_c, sh_c = means.contiguous(), sh.contiguous()
        sr, rr, orw = scaling_raw.contiguous(), rotation_raw.contiguous(), opacity_raw.reshape(-1).contiguous()
        fe = None
        if scaling_modifier == 1.0 and FUSED_FRONTEND:
            # the whole per-Gaussian front end in ONE kernel: activations -> projection -> SH colours -> packed blend records
            # (same values as the separate launches below; an undefined workspace = camera / SH layout not supported)
            # frames of large footprints (lists per 32 x 32 pixels, or more than RANGES_ABOVE list entries per Gaussian: the previous frames' statistics) keep the
            # backward's moment records of a Gaussian in one contiguous run of slots instead of a chain (gsx_raster_common.hpp: "ranges")
            ranges = camera_model == ops.CameraModelType.PINHOLE and _record_ranges_for(_policy_key(width, height, means_c.device.index, means_c.shape[0]))
            fe = ops.frontend_fused_render(sh_degree, means_c, sh_c, sr, rr, orw, viewmat, K, width, height, EPS2D, NEAR_PLANE, FAR_PLANE, RADIUS_CLIP,
                                           camera_model, ut, radial, tangential, None, ranges)
            if fe[8] is None:
                fe = None
        if fe is not None:
            scales, quats, opac, radii, means2d, depths, conics, colors, fe_ws = fe
        elif scaling_modifier == 1.0:   # activations and projection in one launch (bit-identical to the two calls below)
            scales, quats, opac, radii, means2d, depths, conics = ops.splat_activations_projection_ut(
                means_c, sr, rr, orw, viewmat, K, width, height, EPS2D, NEAR_PLANE, FAR_PLANE, RADIUS_CLIP, camera_model, ut, radial, tangential, None)
        else:
            scales, quats, opac = ops.splat_activations_fwd(sr, rr, orw)
            scales = scales * scaling_modifier
            radii, means2d, depths, conics, _ = ops.projection_ut_3dgs_fused(
                means_c, quats, scales, opac, viewmat, None, K, width, height, EPS2D, NEAR_PLANE, FAR_PLANE, RADIUS_CLIP, False,
                camera_model, ut, ops.ShutterType.GLOBAL, radial, tangential, None)
        if fe is None:
            colors = ops.sh_colors_fwd(sh_degree, means_c, viewmat, sh_c, radii)
        # tile size of the intersection lists: 32-px lists only on the fast blend path (global-shutter pinhole) — see _LIST_TILE_STATE
        # (keyed by the image shape, not by the Gaussian count: a model that grows by 5 % every hundred iterations keeps its choice — with N in
        # the key every growth step fell back to 16-px lists for one heavy frame, on a capacity hint from the start of the training)
        lt_key = _policy_key(width, height, means_c.device.index, means_c.shape[0])
        # (and only with the fused front end: its records carry each Gaussian's rectangle of 16-px tiles, which a 16-px tile needs to take
        # exactly the reference's entries out of its 32-px parent's list)
        list_tile = _list_tile_for(lt_key) if (camera_model == ops.CameraModelType.PINHOLE and fe is not None) else TILE_SIZE
        tw, th = (width + list_tile - 1) // list_tile, (height + list_tile - 1) // list_tile
        # binned pipeline: flatten_ids + isect_offsets in one go (bit-identical to intersect_tile + intersect_offset, no isect_ids)
        lists = None
        if guarded and fe is not None:
            # guarded protocol (include/gsx.h): no host read of n_isects — flatten_ids keeps its capacity length, the blend kernels read
            # the frame's verdict on the device, backward() confirms it on the host before anything irreversible
            _, flatten_ids, isect_offsets, lists = ops.intersect_tile_binned_guarded(means2d, radii, depths, 1, list_tile, tw, th)
            if lists.confirmed:   # first call of a problem shape: the exact protocol ran
                _list_tile_update(lt_key, list_tile, int(lists.confirm()[0]), tw * th, means_c.shape[0])
        else:
            _, _, flatten_ids, isect_offsets = ops.intersect_tile_binned(means2d, radii, depths, 1, list_tile, tw, th, False)
            _list_tile_update(lt_key, list_tile, int(flatten_ids.shape[0]), tw * th, means_c.shape[0])
        opac2 = opac.unsqueeze(0)
        if fe is not None:   # the records of exactly these inputs are already in fe_ws
            renders, alphas, last_ids = ops.rasterize_fwd_packed(
                means_c, quats, scales, colors, opac2, bg, None, width, height, list_tile, viewmat, None, K, camera_model, ut,
                ops.ShutterType.GLOBAL, radial, tangential, None, isect_offsets, flatten_ids, fe_ws, lists)
            fwd_ws = fe_ws
        else:
            renders, alphas, last_ids, fwd_ws = ops.rasterize_to_pixels_from_world_3dgs_fwd(
                means_c, quats, scales, colors, opac2, bg, None, width, height, list_tile, viewmat, None, K, camera_model, ut,
                ops.ShutterType.GLOBAL, radial, tangential, None, isect_offsets, flatten_ids, keep_ws=True)
        ctx.fwd_ws = fwd_ws  # packed per-Gaussian records of exactly these inputs: the backward does not pack again
        ctx.save_for_backward(means_c, sh_c, sr, rr, orw, scales, quats, opac2, colors, radii, viewmat, K, isect_offsets,
                              flatten_ids, alphas, last_ids)
        ctx.extra = (bg, width, height, sh_degree, scaling_modifier, camera_model, radial, tangential, grad_sinks, ut)
        ctx.list_tile = list_tile
        ctx.lists = lists
        ctx.lt_update = (lt_key, list_tile, tw * th, means_c.shape[0])
        _TLS.last_lists = lists   # picked up by rasterize_fused right after apply()
        ctx.mark_non_differentiable(radii, means2d, depths, flatten_ids, isect_offsets)
        ctx.set_materialize_grads(False)  # no zero tensors for the outputs nobody differentiates (six fill launches per step)
        return renders, alphas, radii, means2d, depths, flatten_ids, isect_offsets

    @staticmethod
    def backward(ctx, v_renders, v_alphas, *unused):
        if v_renders is None and v_alphas is None:
            return (None,) * 17
        (means, sh, sr, rr, orw, scales, quats, opac2, colors, radii, viewmat, K, isect_offsets, flatten_ids, alphas,
         last_ids) = ctx.saved_tensors
        bg, width, height, sh_degree, scaling_modifier, camera_model, radial, tangential, sinks, ut = ctx.extra
        if v_renders is None:
            v_renders = torch.zeros(alphas.shape[:-1] + (3,), dtype=alphas.dtype, device=alphas.device)
        s = sinks or {}
        try:
            # scaling / rotation / opacity gradients only need the blend backward: they are finished first, so that a multi-GPU caller can
            # start exchanging them ("_early_ready" callback) while the SH backward — 81 % of the gradient bytes — is still running
            # "_regularisers" = (scale_reg / numel, opacity_reg / numel): the MCMC strategy's two regulariser gradients ride on the same kernel
            reg = s.get("_regularisers") or (0.0, 0.0)
            bwd_args = (means, quats, scales, colors, opac2, bg, None, width, height, ctx.list_tile, viewmat, None, K, camera_model, ut,
                        ops.ShutterType.GLOBAL, radial, tangential, None, isect_offsets, flatten_ids, alphas, last_ids,
                        v_renders.contiguous(), None if v_alphas is None else v_alphas.contiguous())
            if scaling_modifier == 1.0 and ACT_EPILOGUE:
                # round 6: the activation Jacobians are the epilogue of the backward's gather kernel (include/gsx.h ABI 7): the raw-parameter
                # gradients leave where v_quats / v_scales / v_opacities sit in registers — one launch and 124 MB at S-1M less per backward
                v_means, v_colors, g_s, g_r, g_o = ops.rasterize_bwd_act(
                    *bwd_args, fwd_ws=ctx.fwd_ws, lists=ctx.lists, raw=(sr, rr, orw),
                    out=(s.get("scaling_raw"), s.get("rotation_raw"), s.get("opacity_raw")), reg=reg)
            else:
                v_means, v_quats, v_scales, v_colors, v_opac = ops.rasterize_to_pixels_from_world_3dgs_bwd(*bwd_args, fwd_ws=ctx.fwd_ws, lists=ctx.lists)
                if scaling_modifier != 1.0:
                    v_scales = v_scales * scaling_modifier
                g_s, g_r, g_o = ops.splat_activations_bwd(sr, rr, orw, v_scales, v_quats, v_opac.reshape(-1), s.get("scaling_raw"),
                                                         s.get("rotation_raw"), s.get("opacity_raw"), float(reg[0]), float(reg[1]))
            n_is = ok = None
            if ctx.lists is not None and not getattr(ctx, "lists_checked", False):
                # Guarded lists: the one place the host looks at the frame's intersection count — with the forward, the loss and the blend
                # backward (~0.9 ms at S-1M) queued behind the 8-byte copy it waits for, so the stream never drains.  Everything so far only
                # overwrote gradient buffers; what follows (the SH tensor's Adam step inside the SH backward, a gradient exchange) is not
                # repeatable, so an overflowed frame stops here.  Under N ranks the verdict is agreed first ("_lists_agree": every rank
                # repeats the iteration when any rank overflowed — the collectives below stay matched).
                n_is, _, ok = ctx.lists.confirm()
                _list_tile_update(ctx.lt_update[0], ctx.lt_update[1], int(n_is), ctx.lt_update[2], ctx.lt_update[3])
        except Exception:
            # N ranks: the peers are (or will be) waiting for this rank's verdict on the frame's lists — a rank that fails ANYWHERE before it
            # votes (the blend backward, the activation Jacobians, the host read of the count: an OOM or a HIP error surfacing there) must still
            # vote, once, or they wait for the agreement's timeout (distributed.ListsAgreement)
            if ctx.lists is not None and s.get("_lists_agree") is not None and not getattr(ctx, "lists_checked", False):
                ctx.lists_checked = True
                s["_lists_agree"](False)
            raise
        if ok is not None:
            ctx.lists_checked = True
            if s.get("_lists_agree") is not None:
                ok = s["_lists_agree"](ok)
            if not ok:
                raise IsectCapacityMiss("rasterize_fused(guarded=True): intersection lists incomplete (%d intersections, capacity %d): "
                                        "run the iteration again" % (n_is, ctx.lists.capacity))
        if s.get("_early_ready") is not None:
            s["_early_ready"]()
        if s.get("_color_exchange") is not None:
            # multi-GPU: the ranks exchange the 3 colour-gradient floats per Gaussian and every rank runs the SH backward over all
            # cameras of the step; the other gradients are all-reduced meanwhile (distributed.ColorGradExchange)
            v_sh, v_means = s["_color_exchange"].sh_backward(sh_degree, means, sh, colors, v_colors, v_means, s["sh"], s["means"],
                                                             sh_adam=s.get("_sh_adam"))
        elif s.get("_sh_adam") is not None:
            # SH backward fused with the Adam step of the SH tensor (optim.FusedAdam.begin_fused_sh_step): sh and its moments are updated
            # in place by the kernel that produces the gradient; the gradient itself is not written (the "sh" sink keeps its old content)
            v_sh = s.get("sh")
            v_means = ops.sh_colors_bwd_adam(sh_degree, means, viewmat, sh, radii, colors, v_colors, v_means, s.get("means"), *s["_sh_adam"])
        else:
            v_sh, v_means = ops.sh_colors_bwd(sh_degree, means, viewmat, sh, radii, colors, v_colors, v_means, s.get("sh"), s.get("means"))
        v_bg = None
        if bg is not None and ctx.needs_input_grad[7]:
            v_bg = (v_renders * (1.0 - alphas)).float().sum(dim=(-3, -2))
        if sinks:  # gradients already sit in the caller's buffers
            return (None,) * 7 + (v_bg,) + (None,) * 9
        return (v_means, v_sh, g_s, g_r, g_o.reshape(-1, 1), None, None, v_bg) + (None,) * 9


def rasterize_fused(camera: Camera, model: SplatData, bg_color: Optional[torch.Tensor], scaling_modifier: float = 1.0,
                    sh_degree: Optional[int] = None, grad_sinks: Optional[dict] = None,
                    with_visibility: bool = False, guarded: bool = False) -> RenderOutput:
    """Same result as rasterize() (RGB mode) through the fused glue kernels.  `grad_sinks` maps
    {"means","sh","scaling_raw","rotation_raw","opacity_raw"} to preallocated gradient buffers that backward
    overwrites (autograd then sees no gradient for the parameters: use the sinks as `.grad`).
    guarded=True: the render never reads n_isects on the host (include/gsx.h "guarded lists"): the intersection lists are filled into
    a capacity taken from the previous frames of the same shape and the kernels check on the device that it sufficed.  The caller
    must be ready to repeat the frame: backward() raises IsectCapacityMiss before anything irreversible when it did not (a
    forward-only caller asks `out.confirm()`).  Same image and gradients as guarded=False whenever the lists were complete."""
    W, H = int(camera.width), int(camera.height)
    viewmat, K = camera.world_view_transform().contiguous(), camera.K_batched().contiguous()
    sh_degree = model.active_sh_degree if sh_degree is None else sh_degree
    cam_model = camera.camera_model if camera.camera_model is not None else ops.CameraModelType.PINHOLE
    bg = bg_color.reshape(1, -1).to(model.means.device).contiguous() if (bg_color is not None and bg_color.numel() > 0) else None
    radial, tangential = _distortion_args(camera, cam_model, model.means.device)
    _TLS.last_lists = None   # (a forward that raises leaves nothing stale behind)
    renders, alphas, radii, means2d, depths, flatten_ids, isect_offsets = GutRenderFunction.apply(
        model.means, model.sh, model.scaling_raw, model.rotation_raw, model.opacity_raw, viewmat, K, bg, W, H, sh_degree,
        scaling_modifier, cam_model, radial, tangential, grad_sinks, guarded)
    out = RenderOutput()
    out.lists, _TLS.last_lists = getattr(_TLS, "last_lists", None), None
    out.render_hwc = renders  # [1,H,W,3] unclamped: what loss.photometric_loss consumes; out.image is derived on first access
    out.alpha = alphas.squeeze(0).permute(2, 0, 1)
    out.means2d, out.depths = means2d, depths.squeeze(0)
    if with_visibility:  # only the densification strategies read these (three more N-sized kernels)
        out.radii = radii.squeeze(0).max(-1).values
        out.visibility = out.radii > 0
    out.width, out.height = W, H
    if out.lists is None:
        out.n_isects = int(flatten_ids.shape[0])
    out.aux = dict(isect_offsets=isect_offsets, flatten_ids=flatten_ids, radii_full=radii)
    return out
