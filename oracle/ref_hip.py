"""TEST INFRASTRUCTURE ONLY — loader of oracle/_ref/gsplat_ref_hip.so: the reference's OWN gsplat operators (its .cu kernels and
.cpp hosts, unmodified, compiled for gfx950 where they lie by oracle/build_ref_hip.sh; bindings: oracle/ref_hip/ref_bind.cpp).

Only tests/ (and tests/golden generators) may import this module: it is the checker that pins the CPU oracle and the HIP kernels
against the reference's kernels executed on the same MI355X.  The product package never imports it.

Argument conventions of the bound functions = gsplat/Ops.h with the enums as ints:
  camera model: PINHOLE 0, ORTHO 1, FISHEYE 2 (gsplat/Common.h:44-48); shutter: ROLLING_TOP_TO_BOTTOM 0, ROLLING_LEFT_TO_RIGHT 1,
  ROLLING_BOTTOM_TO_TOP 2, ROLLING_RIGHT_TO_LEFT 3, GLOBAL 4 (gsplat/Cameras.h:16-22); `ut` = None (defaults) or the 5-float
  tensor of UnscentedTransformParameters::to_tensor().
"""
import importlib.util
import os

HERE = os.path.dirname(os.path.abspath(__file__))
PINHOLE, ORTHO, FISHEYE = 0, 1, 2
ROLLING_TOP_TO_BOTTOM, ROLLING_LEFT_TO_RIGHT, ROLLING_BOTTOM_TO_TOP, ROLLING_RIGHT_TO_LEFT, GLOBAL = 0, 1, 2, 3, 4

_cache = {}


def path(fast=False):
    return os.path.join(HERE, "_ref", "gsplat_ref_hip_fast.so" if fast else "gsplat_ref_hip.so")


def available(fast=False):
    return os.path.exists(path(fast))


def load(fast=False):
    """The extension module, or None when it has not been built (no /root/reference at build time).
    fast=True: the flavour compiled with the reference's release flag --use_fast_math (gsplat/CMakeLists.txt:75)."""
    if fast in _cache:
        return _cache[fast]
    mod = None
    if available(fast):
        import torch  # noqa: F401  (libtorch / libamdhip64 first)
        name = "gsplat_ref_hip_fast" if fast else "gsplat_ref_hip"
        spec = importlib.util.spec_from_file_location(name, path(fast))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    _cache[fast] = mod
    return mod


def render_chain(ref, means, quats, scales, opacities, sh, sh_degree, viewmat, K, width, height, background, camera_model=PINHOLE,
                 shutter=GLOBAL, viewmats1=None, radial=None, tangential=None, thin_prism=None, eps2d=0.3, near=0.01, far=1e4,
                 radius_clip=0.0, calc_compensations=False, v_render_colors=None, v_render_alphas=None, tile=16):
    """One `--gut` render (+ backward) through the reference's operators in the order of gs::training::rasterize
    (src/training/rasterization/rasterizer.cpp:176-181, 248-329): projection -> SH colours (+0.5, clamp_min 0) -> intersect_tile ->
    intersect_offset -> blend forward (-> blend backward).  All arguments are torch tensors on the GPU; returns a dict of tensors."""
    import torch
    C = viewmat.shape[0]
    radii, means2d, depths, conics, comp = ref.projection_ut_3dgs_fused(means, quats, scales, opacities, viewmat, viewmats1, K, width, height, eps2d, near, far,
                                                                       radius_clip, calc_compensations, camera_model, None, shutter, radial, tangential,
                                                                       thin_prism)
    campos = torch.linalg.inv(viewmat.double())[:, :3, 3].float()
    dirs = means[None] - campos[:, None]
    masks = (radii > 0).all(-1)
    colors = ref.spherical_harmonics_fwd(sh_degree, dirs, sh[None].expand(C, -1, -1, -1).contiguous(), masks)
    colors = torch.clamp_min(colors + 0.5, 0.0)
    tw, th = (width + tile - 1) // tile, (height + tile - 1) // tile
    tpg, isect_ids, flatten_ids = ref.intersect_tile(means2d, radii, depths, None, None, C, tile, tw, th, True)
    offsets = ref.intersect_offset(isect_ids, C, tw, th)
    op = opacities[None].expand(C, -1).contiguous()
    renders, alphas, last_ids = ref.rasterize_to_pixels_from_world_3dgs_fwd(means, quats, scales, colors, op, background, None, width, height, tile, viewmat,
                                                                            viewmats1, K, camera_model, None, shutter, radial, tangential, thin_prism,
                                                                            offsets, flatten_ids)
    out = dict(radii=radii, means2d=means2d, depths=depths, conics=conics, compensations=comp, dirs=dirs, masks=masks, colors=colors,
               tiles_per_gauss=tpg, isect_ids=isect_ids, flatten_ids=flatten_ids, tile_offsets=offsets, renders=renders, alphas=alphas,
               last_ids=last_ids)
    if v_render_colors is not None:
        g = ref.rasterize_to_pixels_from_world_3dgs_bwd(means, quats, scales, colors, op, background, None, width, height, tile, viewmat, viewmats1, K,
                                                        camera_model, None, shutter, radial, tangential, thin_prism, offsets, flatten_ids, alphas,
                                                        last_ids, v_render_colors, v_render_alphas)
        out.update(v_means=g[0], v_quats=g[1], v_scales=g[2], v_colors=g[3], v_opacities=g[4])
    return out


def train_path():
    return os.path.join(HERE, "_ref", "gsplat_ref_train.so")


def load_train():
    """oracle/_ref/gsplat_ref_train.so (oracle/build_ref_train.sh): the reference's fused SSIM kernels behind their autograd wrapper
    (fused_ssim / fusedssim / fusedssim_backward) and its fused Adam step (adam_step); None when not built."""
    if "train" in _cache:
        return _cache["train"]
    mod = None
    if os.path.exists(train_path()):
        import torch  # noqa: F401
        spec = importlib.util.spec_from_file_location("gsplat_ref_train", train_path())
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    _cache["train"] = mod
    return mod
