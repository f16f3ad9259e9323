"""The small scenes / cameras on which the reference's own kernels (oracle/_ref/gsplat_ref_hip.so) are run to produce the golden tensors
of tests/golden/ref_hip/*.npz (generator: tests/golden/gen_ref_hip_golden.py, run on the MI355X box), shared by the generator, the CPU
tests of the oracle against those tensors (tests/test_oracle_ref_hip_golden.py) and the GPU tests (tests/test_gpu_reference_hip.py)."""
import numpy as np
import torch

from oracle import ref_hip


def small_scene(scenes, N=4000, size=128, seed=13, f=90.0, sh_degree=0):
    sc = scenes.scene_small(seed=seed, N=N)
    sc["width"] = sc["height"] = size
    sc["K"] = scenes.intrinsics(f, f, size / 2.0, size / 2.0)
    sc["background"] = torch.tensor([0.05, 0.1, 0.15])
    if sh_degree:
        g = torch.Generator().manual_seed(5)
        sc["sh"] = (torch.rand(N, (sh_degree + 1) ** 2, 3, generator=g) - 0.5) * 0.3
        sc["sh_degree"] = sh_degree
    return sc


def cases(scenes):
    """name -> (scene dict, camera extras)"""
    vm1 = scenes.look_at_viewmat((0.03, -0.02, 0.01), (0.05, 0.0, 3.0))[None].numpy()
    return {
        "pinhole_sh3_comp": (small_scene(scenes, sh_degree=3), dict(calc_compensations=True)),
        "distorted_pinhole": (small_scene(scenes), dict(camera_model=ref_hip.PINHOLE, radial=np.array([[0.05, -0.02, 0.003, 0.0, 0.0, 0.0]], np.float32),
                                                         tangential=np.array([[0.002, -0.001]], np.float32),
                                                         thin_prism=np.array([[0.001, 0.0, -0.001, 0.0]], np.float32))),
        "fisheye": (small_scene(scenes, f=70.0), dict(camera_model=ref_hip.FISHEYE, radial=np.array([[0.02, -0.005, 0.001, 0.0]], np.float32))),
        "rolling_top_to_bottom": (small_scene(scenes), dict(shutter=ref_hip.ROLLING_TOP_TO_BOTTOM, viewmats1=vm1)),
        "rolling_left_to_right": (small_scene(scenes), dict(shutter=ref_hip.ROLLING_LEFT_TO_RIGHT, viewmats1=vm1)),
    }


def upstream_grads(sc, seed=3):
    rng = np.random.default_rng(seed)
    H, W = sc["height"], sc["width"]
    return rng.standard_normal((1, H, W, 3)).astype(np.float32), rng.standard_normal((1, H, W, 1)).astype(np.float32)


GOLDEN_KEYS = ["radii", "means2d", "depths", "conics", "compensations", "colors", "tiles_per_gauss", "isect_ids", "flatten_ids", "tile_offsets", "renders",
               "alphas", "last_ids", "v_means", "v_quats", "v_scales", "v_colors", "v_opacities"]
