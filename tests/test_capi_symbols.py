"""The C-ABI library loads on a CPU-only box and exports every symbol include/gsx.h declares (no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "gsx.h")).read()
    return sorted(set(re.findall(r"\b(gsx_[a-z0-9_]+)\s*\(", src)))


def test_libgsx_exports_every_declared_symbol():
    import torch  # noqa: F401  (brings libamdhip64 into the process, as in production)
    lib = ctypes.CDLL(os.path.join(ROOT, "gaussian-splatting-cuda_amd", "libgsx.so"))
    names = _declared()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), "libgsx.so does not export " + n
    assert lib.gsx_abi_version() == 7   # include/gsx.h GSX_ABI_VERSION (history in the header)
    lib.gsx_last_error.restype = ctypes.c_char_p
    assert isinstance(lib.gsx_last_error(), bytes)


def test_argument_validation_without_gpu():
    import torch  # noqa: F401
    lib = ctypes.CDLL(os.path.join(ROOT, "gaussian-splatting-cuda_amd", "libgsx.so"))
    lib.gsx_last_error.restype = ctypes.c_char_p
    # null pointers / bad degree are rejected before any launch
    rc = lib.gsx_spherical_harmonics_fwd(ctypes.c_uint32(3), ctypes.c_uint32(8), ctypes.c_uint32(16), None, None, None, None, None)
    assert rc == -1 and b"null" in lib.gsx_last_error()
    rc = lib.gsx_projection_ut_3dgs_fused(ctypes.c_uint32(4), None, None, None, None, None, 0, 0, ctypes.c_float(0), ctypes.c_float(0),
                                          ctypes.c_float(0), ctypes.c_float(0), None, None, None, None, None, None, None)
    assert rc == -1
    # n == 0 is a successful no-op (the reference skips the launch)
    assert lib.gsx_spherical_harmonics_fwd(ctypes.c_uint32(3), ctypes.c_uint32(0), ctypes.c_uint32(16), None, None, None, None, None) == 0


def test_guarded_entry_points_validate_their_arguments_without_gpu():
    """ABI 4 (include/gsx.h "guarded lists"): null pointers / missing workspaces are rejected before any launch."""
    import torch  # noqa: F401
    c = ctypes
    lib = c.CDLL(os.path.join(ROOT, "gaussian-splatting-cuda_amd", "libgsx.so"))
    lib.gsx_last_error.restype = c.c_char_p
    rc = lib.gsx_intersect_bin_count_guarded(c.c_uint32(1), c.c_uint32(16), None, None, c.c_uint32(16), c.c_uint32(4), c.c_uint32(4), None, None, None,
                                             None, c.c_size_t(0), c.c_int64(100), c.c_int64(0), None, None)
    assert rc == -1 and b"null" in lib.gsx_last_error()
    rc = lib.gsx_rasterize_to_pixels_from_world_3dgs_fwd_guarded(c.c_uint32(4), c.c_int64(8), None, None, None, None, c.c_uint32(3), None, None, None, c.c_uint32(16),
                                                                 c.c_uint32(16), c.c_uint32(16), None, None, None, None, None, None, None, None, c.c_size_t(0),
                                                                 c.c_int(0), None, c.c_int64(0), None)
    assert rc == -1
    rc = lib.gsx_rasterize_to_pixels_from_world_3dgs_bwd_guarded(c.c_uint32(4), c.c_int64(8), None, None, None, None, c.c_uint32(3), None, None, None, c.c_uint32(16),
                                                                 c.c_uint32(16), c.c_uint32(16), None, None, None, None, None, None, None, None, None, None, None,
                                                                 None, None, None, c.c_size_t(0), None, None, c.c_int64(0), None)
    assert rc == -1


def test_ops_module_imports_and_mirrors_ops_h():
    import gsx  # noqa: F401
    from gsx import ops
    for n in ["spherical_harmonics_fwd", "spherical_harmonics_bwd", "intersect_tile", "intersect_offset",
              "projection_ut_3dgs_fused", "rasterize_to_pixels_from_world_3dgs_fwd", "rasterize_to_pixels_from_world_3dgs_bwd"]:
        assert callable(getattr(ops, n))
    assert int(ops.CameraModelType.FISHEYE) == 2 and int(ops.ShutterType.GLOBAL) == 4
    ut = ops.UnscentedTransformParameters()
    assert abs(ut.alpha - 0.1) < 1e-7 and ut.beta == 2.0 and ut.require_all_sigma_points_valid


def test_argument_validation_of_the_next_tier_entry_points():
    """Null pointers, undersized workspaces and unsupported shapes are rejected before any launch (runs without a GPU)."""
    import torch  # noqa: F401
    c = ctypes
    lib = c.CDLL(os.path.join(ROOT, "gaussian-splatting-cuda_amd", "libgsx.so"))
    lib.gsx_last_error.restype = c.c_char_p
    for f in ("gsx_photometric_loss_workspace_bytes", "gsx_intersect_bin_count_workspace_bytes", "gsx_intersect_bin_fill_workspace_bytes",
              "gsx_rasterize_fwd_workspace_bytes", "gsx_rasterize_bwd_workspace_bytes"):
        getattr(lib, f).restype = c.c_size_t
    lib.gsx_rasterize_fwd_packed_records.restype = c.c_void_p
    u32, f32 = c.c_uint32, c.c_float
    # Adam: n == 0 is a no-op, null pointers / split beyond cols / unaligned cols are errors
    assert lib.gsx_adam_step(c.c_uint64(0), u32(3), c.c_uint64(3), c.c_uint64(3), None, None, None, None, f32(0), f32(0), f32(0), f32(0), f32(1), f32(1), None) == 0
    assert lib.gsx_adam_step(c.c_uint64(4), u32(3), c.c_uint64(3), c.c_uint64(3), None, None, None, None, f32(0), f32(0), f32(0), f32(0), f32(1), f32(1), None) == -1
    buf = (c.c_float * 64)()
    assert lib.gsx_adam_step_split(c.c_uint64(2), u32(12), u32(13), buf, buf, buf, buf, f32(0), f32(0), 1, 1, f32(0), f32(0), f32(0), f32(1), f32(1), None) == -1
    assert lib.gsx_adam_step_split(c.c_uint64(2), u32(9), u32(3), buf, buf, buf, buf, f32(0), f32(0), 1, 1, f32(0), f32(0), f32(0), f32(1), f32(1), None) == -2
    # SSIM / loss
    assert lib.gsx_fused_ssim_fwd(u32(0), u32(3), u32(8), u32(8), f32(1e-4), f32(9e-4), None, None, None, None, None, None, None) == 0
    assert lib.gsx_fused_ssim_fwd(u32(1), u32(3), u32(8), u32(8), f32(1e-4), f32(9e-4), None, None, None, None, None, None, None) == -1
    assert lib.gsx_fused_ssim_fwd(u32(1), u32(3), u32(8), u32(8), f32(1e-4), f32(9e-4), buf, buf, buf, buf, None, None, None) == -1   # 1 of 3 maps
    need = lib.gsx_photometric_loss_workspace_bytes(u32(1), u32(32), u32(48))
    assert need >= 9 * 32 * 48 * 4
    assert lib.gsx_photometric_loss_fwd(u32(1), u32(32), u32(48), f32(0.2), buf, buf, buf, buf, c.c_size_t(16), None) == -3
    assert b"workspace" in lib.gsx_last_error()
    assert lib.gsx_photometric_loss_bwd(u32(1), u32(32), u32(48), f32(0.2), None, f32(1), buf, buf, None, c.c_size_t(need), buf, None) == -1
    # binned intersection: tile grid limit, workspace, empty input
    assert lib.gsx_intersect_bin_supported(u32(120), u32(68)) == 1 and lib.gsx_intersect_bin_supported(u32(480), u32(270)) == 0
    off = (c.c_int32 * 16)()
    assert lib.gsx_intersect_bin_count(u32(1), u32(8), buf, off, u32(16), u32(480), u32(270), None, off, None, buf, c.c_size_t(1 << 30), None) == -2
    assert lib.gsx_intersect_bin_count(u32(1), u32(8), buf, off, u32(16), u32(2), u32(2), None, off, None, buf, c.c_size_t(8), None) == -3
    assert lib.gsx_intersect_bin_fill(u32(1), u32(8), buf, off, buf, u32(16), u32(2), u32(2), off, c.c_int64(0), c.c_int64(0), buf, off, None, None, c.c_size_t(0), None) == 0
    assert lib.gsx_intersect_bin_fill_workspace_bytes(u32(1), u32(2), u32(2), c.c_int64(100)) >= 2 * 100 * 8
    # ranked variant: bitmap capacity, argument checks (no launch happens without valid pointers)
    assert lib.gsx_intersect_ranked_supported(u32(1), u32(1 << 20)) == 1 and lib.gsx_intersect_ranked_supported(u32(2), u32(1 << 19)) == 1
    assert lib.gsx_intersect_ranked_supported(u32(1), u32((1 << 20) + 1)) == 0 and lib.gsx_intersect_ranked_supported(u32(0), u32(5)) == 0
    assert lib.gsx_intersect_depth_ranks_workspace_bytes(u32(1), u32(1000)) >= 3 * 1000 * 4
    assert lib.gsx_intersect_depth_ranks(u32(1), u32((1 << 20) + 1), off, buf, off, off, buf, c.c_size_t(1 << 30), None) == -2
    assert lib.gsx_intersect_depth_ranks(u32(1), u32(8), None, buf, off, off, buf, c.c_size_t(1 << 30), None) == -1
    assert lib.gsx_intersect_depth_ranks(u32(1), u32(8), off, buf, off, off, buf, c.c_size_t(8), None) == -3
    assert lib.gsx_intersect_bin_fill_ranked_workspace_bytes(c.c_int64(100)) >= 100 * 4
    assert lib.gsx_intersect_bin_fill_ranked(u32(1), u32(8), buf, off, buf, u32(16), u32(2), u32(2), off, c.c_int64(0), buf, off, off, off, None, None,
                                             c.c_size_t(0), None) == 0     # nothing to fill
    assert lib.gsx_intersect_bin_fill_ranked(u32(1), u32(8), buf, off, buf, u32(16), u32(2), u32(2), off, c.c_int64(5), buf, None, off, off, None, buf,
                                             c.c_size_t(1 << 20), None) == -1    # ranks missing
    # blend workspaces
    assert lib.gsx_rasterize_fwd_packed_records(None, c.c_size_t(0), u32(1), u32(8)) is None
    assert lib.gsx_rasterize_bwd_workspace_bytes(u32(1), u32(1000), c.c_int64(5000)) >= 5000 * 64 + 1000 * 4 + 1000 * 64


def test_shim_exports_the_reference_operator_symbols():
    """The C++ shim defines, with the reference's exact (mangled) signatures, every operator the `--gut` training step links
    against: the ten gsplat:: functions of Ops.h, fast_gs::optimizer::adam_step(_wrapper) and fusedssim(_backward)."""
    import glob
    import subprocess
    so = glob.glob(os.path.join(ROOT, "gaussian-splatting-cuda_amd", "_gsx_ops*.so"))[0]
    syms = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    for needle in ["_ZN6gsplat23spherical_harmonics_fwd", "_ZN6gsplat23spherical_harmonics_bwd", "_ZN6gsplat14intersect_tile",
                   "_ZN6gsplat16intersect_offset", "_ZN6gsplat24projection_ut_3dgs_fused",
                   "_ZN6gsplat39rasterize_to_pixels_from_world_3dgs_fwd", "_ZN6gsplat39rasterize_to_pixels_from_world_3dgs_bwd",
                   "_ZN6gsplat16quats_to_rotmats", "_ZN6gsplat10relocation", "_ZN6gsplat9add_noise",
                   "_ZN7fast_gs9optimizer17adam_step_wrapperERN2at6TensorES3_S3_RKS2_ffffff",
                   "_ZN7fast_gs9optimizer9adam_stepEPfS1_S1_PKfiffffff", "_Z9fusedssimffRN2at6TensorES1_b",
                   "_Z18fusedssim_backwardffRN2at6TensorES1_S1_S1_S1_S1_"]:
        assert needle in syms, needle


def test_ab_switches_are_gated():
    """libgsx.so looks at its A/B environment switches only when GSX_TEST_SWITCHES=1 (an embedding host inherits none of them):
    include/gsx.h gsx_test_switch.  The gate is read once per process, hence the subprocesses."""
    import subprocess
    import sys
    code = ("import ctypes,sys; lib=ctypes.CDLL(sys.argv[1]); lib.gsx_test_switch.restype=ctypes.c_char_p; "
            "print(lib.gsx_test_switch(b'GSX_BWD'))")
    lib_path = os.path.join(ROOT, "gaussian-splatting-cuda_amd", "libgsx.so")
    env = {k: v for k, v in os.environ.items() if k != "GSX_TEST_SWITCHES"}
    env["GSX_BWD"] = "pm"
    off = subprocess.run([sys.executable, "-c", code, lib_path], env=env, stdout=subprocess.PIPE, text=True, check=True).stdout.strip()
    on = subprocess.run([sys.executable, "-c", code, lib_path], env=dict(env, GSX_TEST_SWITCHES="1"), stdout=subprocess.PIPE, text=True, check=True).stdout.strip()
    assert off == "None" and on == "b'pm'"
