"""Blender / NeRF-synthetic `transforms*.json` reader, following src/loader/formats/transforms.cpp:
camera-to-world matrices with the OpenGL axis convention -> world-to-camera [R|t] (flip the y / z columns, invert, rotate by pi
about y: transforms.cpp:196-216), focal length from fl_x / fl_y or camera_angle_x / camera_angle_y (:137-155), principal point
cx / cy or the image centre (:157-168), image size from w / h or from the first image (:103-131), non-zero distortion rejected
(:170-190), `<file_path>.png` preferred when it exists (:52-60), and the random initial point cloud used when a scene ships
no points (:248-262: 10 000 points in [-1,1]^3, torch seed 8128)."""
import json
import math
import os

import numpy as np
import torch

from . import ops
from .io_colmap import ColmapCamera, ColmapScene
from .rasterizer import Camera

DEFAULT_NUM_INIT_GAUSSIAN = 10000
DEFAULT_RANDOM_SEED = 8128


def fov_rad_to_focal_length(resolution, fov_rad):
    return 0.5 * float(resolution) / math.tan(0.5 * fov_rad)


def _image_path(dir_path, frame):
    p = os.path.join(dir_path, frame["file_path"])
    return p + ".png" if os.path.exists(p + ".png") else p


def load_transforms(path, device="cpu"):
    tf = path
    if os.path.isdir(path):
        for name in ("transforms_train.json", "transforms.json"):
            if os.path.isfile(os.path.join(path, name)):
                tf = os.path.join(path, name)
                break
        else:
            raise RuntimeError("could not find transforms_train.json nor transforms.json in " + path)
    if not os.path.isfile(tf):
        raise RuntimeError(tf + " is not a valid file")
    dir_path = os.path.dirname(tf)
    with open(tf, "r") as f:
        t = json.load(f)
    if "w" in t and "h" in t:
        w, h = int(t["w"]), int(t["h"])
    else:
        try:
            from PIL import Image
            with Image.open(_image_path(dir_path, t["frames"][0])) as im:
                w, h = im.size
        except Exception as e:  # noqa: BLE001
            raise RuntimeError("Error while trying to read image dimensions: " + str(e)) from e
    fl_x = float(t["fl_x"]) if "fl_x" in t else (fov_rad_to_focal_length(w, float(t["camera_angle_x"])) if "camera_angle_x" in t else -1.0)
    if "fl_y" in t:
        fl_y = float(t["fl_y"])
    elif "camera_angle_y" in t:
        fl_y = fov_rad_to_focal_length(h, float(t["camera_angle_y"]))
    else:
        if w != h:
            raise RuntimeError("no camera_angle_y expected w!=h")
        fl_y = fl_x
    cx, cy = float(t.get("cx", 0.5 * w)), float(t.get("cy", 0.5 * h))
    k1, k2, p1, p2 = (float(t.get(k, 0.0)) for k in ("k1", "k2", "p1", "p2"))
    if k1 > 0 or k2 > 0 or p1 > 0 or p2 > 0:
        raise RuntimeError(f"GS don't support distortion for now: k1={k1}, k2={k2}, p1={p1}, p2={p2}")
    fix = np.eye(4, dtype=np.float32)   # rotation by pi about y (transforms.cpp:31-48)
    fix[0, 0], fix[0, 2], fix[2, 0], fix[2, 2] = math.cos(math.pi), math.sin(math.pi), -math.sin(math.pi), math.cos(math.pi)
    scene = ColmapScene()
    locs = []
    for uid, frame in enumerate(t.get("frames", [])):
        if "transform_matrix" not in frame:   # transforms.cpp:187-195 throws; skipping would also shift the uid <-> image correspondence
            raise RuntimeError("expected all frames to contain transform_matrix")
        c2w = np.array(frame["transform_matrix"], np.float32)
        if c2w.shape != (4, 4):
            raise RuntimeError("transform_matrix has the wrong dimensions")
        c2w = c2w.copy()
        c2w[0:3, 1:3] *= -1.0
        w2c = np.linalg.inv(c2w.astype(np.float64)).astype(np.float32) @ fix
        vm = np.eye(4, dtype=np.float32)
        vm[:3, :3], vm[:3, 3] = w2c[:3, :3], w2c[:3, 3]
        locs.append(-vm[:3, :3].T @ vm[:3, 3])
        K = np.array([[fl_x, 0, cx], [0, fl_y, cy], [0, 0, 1]], np.float32)
        ip = _image_path(dir_path, frame)
        cam = Camera(viewmat=torch.from_numpy(vm).to(device), K=torch.from_numpy(K).to(device), width=w, height=h,
                     camera_model=ops.CameraModelType.PINHOLE)
        scene.cameras.append(ColmapCamera(cam, os.path.basename(ip), ip, uid, "PINHOLE"))
    scene.camera_locations = np.stack(locs).astype(np.float32) if locs else np.zeros((0, 3), np.float32)
    scene.scene_center = np.zeros(3, np.float32)   # transforms.cpp:238
    return scene


def generate_random_point_cloud():
    """transforms.cpp:248-262; positions float32 [10000,3] in [-1,1], colours uint8."""
    g = torch.Generator().manual_seed(DEFAULT_RANDOM_SEED)
    positions = torch.rand(DEFAULT_NUM_INIT_GAUSSIAN, 3, generator=g) * 2.0 - 1.0
    colors = torch.randint(0, 256, (DEFAULT_NUM_INIT_GAUSSIAN, 3), generator=g, dtype=torch.uint8)
    return positions.numpy(), colors.numpy()
