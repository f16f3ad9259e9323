"""The gsplat operator surface (gsplat/Ops.h:12-43,69-166 of the reference) on the HIP backend.

Everything here is the C++ shim `_gsx_ops` (csrc/ops_shim.cpp) over the C ABI of libgsx.so.  There is NO
fallback: if the native module is missing or cannot be loaded this import fails loudly.
"""
import os
import sys

import torch  # noqa: F401  (loads libamdhip64 / libtorch before the extension)

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)
try:
    import _gsx_ops as _C
except ImportError as e:  # pragma: no cover
    raise ImportError(
        "gsx: the native HIP extension (_gsx_ops / libgsx.so) is not built or failed to load; run "
        "`python __graft_entry__.py` (build()) first. There is no CPU or PyTorch fallback. Cause: %s" % e)

CameraModelType = _C.CameraModelType
ShutterType = _C.ShutterType
UnscentedTransformParameters = _C.UnscentedTransformParameters

spherical_harmonics_fwd = _C.spherical_harmonics_fwd
spherical_harmonics_bwd = _C.spherical_harmonics_bwd
intersect_tile = _C.intersect_tile
intersect_offset = _C.intersect_offset
projection_ut_3dgs_fused = _C.projection_ut_3dgs_fused


def rasterize_to_pixels_from_world_3dgs_fwd(*args, keep_ws=False):
    """gsplat::rasterize_to_pixels_from_world_3dgs_fwd; keep_ws=True also returns the workspace (packed per-Gaussian records)
    that the backward of the same inputs can take back (fwd_ws=) instead of re-packing."""
    return _C.rasterize_fwd_keep_ws(*args) if keep_ws else _C.rasterize_to_pixels_from_world_3dgs_fwd(*args)


def rasterize_to_pixels_from_world_3dgs_bwd(*args, fwd_ws=None, lists=None):
    """`lists`: the IsectLists handle of intersect_tile_binned_guarded the forward ran with (guarded protocol, include/gsx.h)."""
    return _C.rasterize_to_pixels_from_world_3dgs_bwd(*args, fwd_ws, lists)


def rasterize_bwd_act(*args, fwd_ws=None, lists=None, raw=None, out=(None, None, None), reg=(0.0, 0.0)):
    """The blend backward of ONE camera through to the raw SplatData parameters (include/gsx.h, ABI 7): `args` as for
    rasterize_to_pixels_from_world_3dgs_bwd, raw = (scaling_raw, rotation_raw, opacity_raw [N]), out = optional gradient buffers, reg = the
    regulariser gradients per element.  Returns (v_means, v_colors, v_scaling_raw, v_rotation_raw, v_opacity_raw)."""
    return _C.rasterize_bwd_act(*args, fwd_ws, lists, raw[0], raw[1], raw[2], out[0], out[1], out[2], float(reg[0]), float(reg[1]))


quats_to_rotmats = _C.quats_to_rotmats
relocation = _C.relocation
add_noise = _C.add_noise
abi_version = _C.abi_version

# fused glue ops (extensions beyond Ops.h; include/gsx.h "fused glue")
sh_colors_fwd = _C.sh_colors_fwd
sh_colors_bwd = _C.sh_colors_bwd
sh_colors_bwd_adam = _C.sh_colors_bwd_adam
splat_activations_fwd = _C.splat_activations_fwd
splat_activations_projection_ut = _C.splat_activations_projection_ut
frontend_fused_render = _C.frontend_fused_render   # the same launch without the conics output (rasterize_fused)
frontend_fused = _C.frontend_fused            # activations -> UT projection -> SH colours -> packed records, one kernel (gsx_frontend.hip)
rasterize_fwd_packed = _C.rasterize_fwd_packed
splat_activations_bwd = _C.splat_activations_bwd
adam_step = _C.adam_step
adam_step_multi = _C.adam_step_multi
adam_step_wrapper = _C.adam_step_wrapper
intersect_tile_binned = _C.intersect_tile_binned
intersect_tile_binned_guarded = _C.intersect_tile_binned_guarded   # no host read of n_isects: (tiles_per_gauss, flatten_ids [capacity], offsets, IsectLists)
shim_guarded_stats = _C.shim_guarded_stats
intersect_tile_device_sort = _C.intersect_tile_device_sort
adam_step_split = _C.adam_step_split
fusedssim = _C.fusedssim
fusedssim_backward = _C.fusedssim_backward
photometric_loss_fwd = _C.photometric_loss_fwd
photometric_loss_bwd = _C.photometric_loss_bwd
photometric_loss_single_pass = _C.photometric_loss_single_pass   # loss3 and v_render in one kernel (include/gsx.h ABI 7)
shim_stats = _C.shim_stats
shim_ranked_calls = _C.shim_ranked_calls
