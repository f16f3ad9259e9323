"""COLMAP sparse-model reader: cameras.bin / images.bin / points3D.bin -> rasterizer.Camera list +
initial point cloud, following src/loader/formats/colmap.cpp (binary layouts :300-470, per-model intrinsics :684-840,
world->camera [R|t] from (qvec, tvec) :25-51, camera centres = -R^T t :676) and the model initialisation of
SplatData::init_model_from_pointcloud (src/core/splat_data.cpp:63-111, 506-600).

Everything the rasterizer needs (pose, intrinsics, distortion, size) comes from the sparse model (.bin, or .txt when the
binary files are absent); `ColmapScene.load_images()` decodes the image files through io_image."""
import os
import struct
from dataclasses import dataclass, field
from typing import List

import numpy as np
import torch

from . import ops
from .rasterizer import Camera, SplatData

# model id -> (name, number of parameters)   (colmap.cpp:117-129)
CAMERA_MODELS = {0: ("SIMPLE_PINHOLE", 3), 1: ("PINHOLE", 4), 2: ("SIMPLE_RADIAL", 4), 3: ("RADIAL", 5), 4: ("OPENCV", 8),
                 5: ("OPENCV_FISHEYE", 8), 6: ("FULL_OPENCV", 12), 7: ("FOV", 5), 8: ("SIMPLE_RADIAL_FISHEYE", 4), 9: ("RADIAL_FISHEYE", 5),
                 10: ("THIN_PRISM_FISHEYE", 12)}


def qvec2rotmat(q):
    """colmap.cpp:30-51 (w, x, y, z): the quaternion is normalised (torch F::normalize, eps 1e-12) and R composed in fp32, operation by
    operation as there — R is bit-identical to the reference reader's (tests/test_io.py checks it against that reader)."""
    qn = torch.nn.functional.normalize(torch.as_tensor(np.asarray(q, dtype=np.float32)), dim=0).numpy()
    w, x, y, z = (np.float32(v) for v in qn)
    one, two = np.float32(1.0), np.float32(2.0)
    return np.array([[one - two * (y * y + z * z), two * (x * y - z * w), two * (x * z + y * w)],
                     [two * (x * y + z * w), one - two * (x * x + z * z), two * (y * z - x * w)],
                     [two * (x * z - y * w), two * (y * z + x * w), one - two * (x * x + y * y)]], np.float32)


def read_cameras_binary(path, scale_factor=1.0):
    buf = open(path, "rb").read()
    (n,), off, cams = struct.unpack_from("<Q", buf, 0), 8, {}
    for _ in range(n):
        cam_id, model_id, w, h = struct.unpack_from("<IiQQ", buf, off)
        off += 24
        if model_id not in CAMERA_MODELS:
            raise RuntimeError(f"Unsupported camera-model id {model_id}")
        name, cnt = CAMERA_MODELS[model_id]
        params = list(struct.unpack_from("<%dd" % cnt, buf, off))
        off += 8 * cnt
        if scale_factor != 1.0:  # images_<k> folders: colmap.cpp:172-258, 365-383
            w, h = int(w / scale_factor), int(h / scale_factor)
            n_focal = 1 if name in ("SIMPLE_PINHOLE", "SIMPLE_RADIAL", "RADIAL", "SIMPLE_RADIAL_FISHEYE", "RADIAL_FISHEYE") else 2
            for k in range(n_focal + 2):
                params[k] /= scale_factor
        cams[cam_id] = {"model": name, "width": int(w), "height": int(h), "params": np.array(params, np.float64).astype(np.float32)}
    if off != len(buf):
        raise RuntimeError("cameras.bin: trailing bytes")
    return cams


def read_images_binary(path):
    buf = open(path, "rb").read()
    (n,), off, images = struct.unpack_from("<Q", buf, 0), 8, []
    for _ in range(n):
        image_id, = struct.unpack_from("<I", buf, off)
        q = np.array(struct.unpack_from("<4d", buf, off + 4), np.float64).astype(np.float32)
        t = np.array(struct.unpack_from("<3d", buf, off + 36), np.float64).astype(np.float32)
        cam_id, = struct.unpack_from("<I", buf, off + 60)
        off += 64
        end = buf.index(b"\0", off)
        name = buf[off:end].decode("utf-8")
        off = end + 1
        npts, = struct.unpack_from("<Q", buf, off)
        off += 8 + npts * 24
        images.append({"id": image_id, "qvec": q, "tvec": t, "camera_id": cam_id, "name": name})
    if off != len(buf):
        raise RuntimeError("images.bin: trailing bytes")
    return images


def read_points3D_binary(path):
    buf = open(path, "rb").read()
    (n,), off = struct.unpack_from("<Q", buf, 0), 8
    xyz, rgb = np.empty((n, 3), np.float32), np.empty((n, 3), np.uint8)
    for i in range(n):
        x, y, z = struct.unpack_from("<3d", buf, off + 8)
        xyz[i] = (x, y, z)
        rgb[i] = struct.unpack_from("<3B", buf, off + 32)
        track_len, = struct.unpack_from("<Q", buf, off + 43)
        off += 51 + 8 * track_len
    if off != len(buf):
        raise RuntimeError("points3D.bin: trailing bytes")
    return xyz, rgb


MODEL_IDS = {name: (mid, cnt) for mid, (name, cnt) in CAMERA_MODELS.items()}


def _data_lines(path):
    with open(path, "r") as f:
        return [ln.strip() for ln in f if ln.strip() and not ln.lstrip().startswith("#")]


def _scale_params(name, params, w, h, scale_factor):
    if scale_factor == 1.0:
        return params, w, h
    n_focal = 1 if name in ("SIMPLE_PINHOLE", "SIMPLE_RADIAL", "RADIAL", "SIMPLE_RADIAL_FISHEYE", "RADIAL_FISHEYE") else 2
    params = list(params)
    for k in range(n_focal + 2):
        params[k] /= scale_factor
    return params, int(w / scale_factor), int(h / scale_factor)


def read_cameras_text(path, scale_factor=1.0):
    """cameras.txt: CAMERA_ID MODEL WIDTH HEIGHT PARAMS[] (colmap.cpp: text readers)."""
    cams = {}
    for ln in _data_lines(path):
        tok = ln.split()
        cam_id, name, w, h = int(tok[0]), tok[1], int(tok[2]), int(tok[3])
        if name not in MODEL_IDS:
            raise RuntimeError(f"Unsupported camera model {name}")
        params = [float(v) for v in tok[4:4 + MODEL_IDS[name][1]]]
        params, w, h = _scale_params(name, params, w, h, scale_factor)
        cams[cam_id] = {"model": name, "width": w, "height": h, "params": np.array(params, np.float64).astype(np.float32)}
    return cams


def read_images_text(path):
    """images.txt: two lines per image (IMAGE_ID QW QX QY QZ TX TY TZ CAMERA_ID NAME / POINTS2D[])."""
    images = []
    with open(path, "r") as f:
        lines = [ln.rstrip("\n") for ln in f if not ln.lstrip().startswith("#")]
    i = 0
    while i < len(lines):
        if not lines[i].strip():
            i += 1
            continue
        tok = lines[i].split()
        images.append({"id": int(tok[0]), "qvec": np.array([float(v) for v in tok[1:5]], np.float64).astype(np.float32),
                       "tvec": np.array([float(v) for v in tok[5:8]], np.float64).astype(np.float32), "camera_id": int(tok[8]),
                       "name": " ".join(tok[9:])})
        i += 2  # skip the 2-D observations line (possibly empty)
    return images


def read_points3D_text(path):
    """points3D.txt: POINT3D_ID X Y Z R G B ERROR TRACK[] (colmap.cpp:612-645)."""
    rows = [ln.split() for ln in _data_lines(path)]
    for r in rows:
        if len(r) < 8:
            raise RuntimeError("Invalid format in point3D.txt: " + " ".join(r))
    xyz = np.array([[float(r[1]), float(r[2]), float(r[3])] for r in rows], np.float32).reshape(-1, 3)
    rgb = np.array([[int(r[4]), int(r[5]), int(r[6])] for r in rows], np.uint8).reshape(-1, 3)
    return xyz, rgb


@dataclass
class ColmapCamera:
    camera: Camera
    image_name: str
    image_path: str
    uid: int
    model: str


@dataclass
class ColmapScene:
    """`load_images(device)` decodes every camera's image file (io_image) into float [3,H,W] tensors, in camera order."""
    cameras: List[ColmapCamera] = field(default_factory=list)
    camera_locations: np.ndarray = None   # [n,3] world-space centres
    scene_center: np.ndarray = None       # their mean (colmap.cpp:  scene centre used for the scene scale)
    points: np.ndarray = None             # [P,3] float32
    colors: np.ndarray = None             # [P,3] uint8

    def load_images(self, device="cpu", res_div=1, max_width=0):
        from . import io_image
        return [io_image.load_and_get_image(c.image_path, res_div, max_width, device) for c in self.cameras]


def _intrinsics(model, p):
    """colmap.cpp:684-840 -> (fx, fy, cx, cy, radial, tangential, gsplat camera model)."""
    PIN, FISH = ops.CameraModelType.PINHOLE, ops.CameraModelType.FISHEYE
    if model == "SIMPLE_PINHOLE":
        return p[0], p[0], p[1], p[2], None, None, PIN
    if model == "PINHOLE":
        return p[0], p[1], p[2], p[3], None, None, PIN
    if model == "SIMPLE_RADIAL":
        return p[0], p[0], p[1], p[2], (np.array([p[3]], np.float32) if p[3] != 0 else None), None, PIN
    if model == "RADIAL":
        return p[0], p[0], p[1], p[2], p[3:5].copy(), None, PIN
    if model == "OPENCV":
        return p[0], p[1], p[2], p[3], p[4:6].copy(), p[6:8].copy(), PIN
    if model == "FULL_OPENCV":
        return p[0], p[1], p[2], p[3], np.array([p[4], p[5], p[8], p[9], p[10], p[11]], np.float32), p[6:8].copy(), PIN
    if model == "OPENCV_FISHEYE":
        return p[0], p[1], p[2], p[3], p[4:8].copy(), None, FISH
    if model == "RADIAL_FISHEYE":
        return p[0], p[0], p[1], p[2], p[3:5].copy(), None, FISH
    if model == "SIMPLE_RADIAL_FISHEYE":
        return p[0], p[0], p[1], p[2], p[3:4].copy(), None, FISH
    if model == "THIN_PRISM_FISHEYE":
        raise RuntimeError("THIN_PRISM_FISHEYE camera model is not supported but could be implemented in 3DGUT pretty easily")
    if model == "FOV":
        raise RuntimeError("FOV camera model is not supported.")
    raise RuntimeError("Unsupported camera model")


def _pad4(a):
    """rasterizer.cpp:183-195 pads distortion vectors to >= 4 entries."""
    if a is None:
        return None
    out = np.zeros(max(4, len(a)), np.float32)
    out[:len(a)] = a
    return out


def _correct_dimensions(cameras):
    """colmap.cpp:852-877: when the FIRST image file exists and its size differs from the COLMAP database's (relative difference above
    1e-5 on either axis), every camera takes that file's size and has fx, cx scaled by actual_w / expected_w and fy, cy by
    actual_h / expected_h of the first camera (fp32 arithmetic as there); distortion coefficients are dimensionless and stay."""
    if not cameras or not os.path.exists(cameras[0].image_path):
        return
    from PIL import Image
    try:
        with Image.open(cameras[0].image_path) as im:
            actual_w, actual_h = im.size
    except Exception as e:  # noqa: BLE001
        raise RuntimeError(f"Load failed: {cameras[0].image_path} : {e}") from e
    c0 = cameras[0].camera
    sx, sy = np.float32(actual_w) / np.float32(c0.width), np.float32(actual_h) / np.float32(c0.height)
    if abs(sx - np.float32(1.0)) <= 1e-5 and abs(sy - np.float32(1.0)) <= 1e-5:
        return
    scale = torch.tensor([[sx, 1.0, sx], [1.0, sy, sy], [1.0, 1.0, 1.0]], dtype=torch.float32)
    for c in cameras:
        c.camera.width, c.camera.height = int(actual_w), int(actual_h)
        c.camera.K = c.camera.K * scale.to(c.camera.K.device)


def load_colmap(base_path, images_folder="images", device="cpu"):
    sparse = os.path.join(base_path, "sparse", "0")
    if not os.path.isdir(sparse):
        sparse = os.path.join(base_path, "sparse")
    suffix = images_folder.rsplit("_", 1)[-1] if "_" in images_folder else ""
    try:
        factor = float(suffix)
        factor = factor if 0 < factor <= 16 else 1.0
    except ValueError:
        factor = 1.0
    def pick(stem):
        b, t = os.path.join(sparse, stem + ".bin"), os.path.join(sparse, stem + ".txt")
        return (b, True) if os.path.exists(b) else (t, False)
    cpath, cbin = pick("cameras")
    ipath, ibin = pick("images")
    if not os.path.exists(cpath) or not os.path.exists(ipath):
        raise RuntimeError(f"COLMAP sparse model not found under {sparse} (cameras / images .bin or .txt)")
    cams = read_cameras_binary(cpath, factor) if cbin else read_cameras_text(cpath, factor)
    images = read_images_binary(ipath) if ibin else read_images_text(ipath)
    scene = ColmapScene()
    locs = []
    for uid, img in enumerate(images):
        if img["camera_id"] not in cams:
            raise RuntimeError(f"Camera ID {img['camera_id']} not found")
        c = cams[img["camera_id"]]
        R, t = qvec2rotmat(img["qvec"]), img["tvec"]
        locs.append(-R.T @ t)
        fx, fy, cx, cy, radial, tangential, kind = _intrinsics(c["model"], c["params"])
        vm = np.eye(4, dtype=np.float32)
        vm[:3, :3], vm[:3, 3] = R, t
        K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], np.float32)
        radial, tangential = _pad4(radial), _pad4(tangential)
        cam = Camera(viewmat=torch.from_numpy(vm).to(device), K=torch.from_numpy(K).to(device), width=c["width"], height=c["height"],
                     camera_model=kind, radial=None if radial is None else torch.from_numpy(radial).to(device),
                     tangential=None if tangential is None else torch.from_numpy(tangential).to(device))
        scene.cameras.append(ColmapCamera(cam, img["name"], os.path.join(base_path, images_folder, img["name"]), uid, c["model"]))
    _correct_dimensions(scene.cameras)
    scene.camera_locations = np.stack(locs) if locs else np.zeros((0, 3), np.float32)
    scene.scene_center = scene.camera_locations.mean(0) if locs else np.zeros(3, np.float32)
    p3d, p3d_txt = os.path.join(sparse, "points3D.bin"), os.path.join(sparse, "points3D.txt")
    if os.path.exists(p3d):
        scene.points, scene.colors = read_points3D_binary(p3d)
    elif os.path.exists(p3d_txt):
        scene.points, scene.colors = read_points3D_text(p3d_txt)
    return scene


def mean_neighbor_distances(points):
    """compute_mean_neighbor_distances (splat_data.cpp:63-111): mean distance to the (up to) 3 nearest neighbours farther
    than 1e-4 (squared 1e-8) among the 4 nearest results, 0.01 when there is none."""
    from scipy.spatial import cKDTree
    n = points.shape[0]
    if n <= 1:
        return np.full(n, 0.01, np.float32)
    k = min(4, n)
    d, _ = cKDTree(points).query(points, k=k)
    d = d.reshape(n, k).astype(np.float32)
    out = np.empty(n, np.float32)
    for i in range(n):
        valid = d[i][(d[i] * d[i]) > 1e-8][:3]
        out[i] = valid.mean() if valid.size else 0.01
    return out


def init_model_from_pointcloud(points, colors_u8, scene_center, sh_degree=3, init_scaling=0.1, init_opacity=0.5, device="cpu"):
    """SplatData::init_model_from_pointcloud (splat_data.cpp:506-600), non-random branch.  Returns (SplatData, scene_scale)."""
    pts = np.ascontiguousarray(points, np.float32)
    cols = colors_u8.astype(np.float32) / 255.0
    scene_scale = float(np.median(np.linalg.norm(pts - np.asarray(scene_center, np.float32)[None], axis=1)))
    nn = np.maximum(mean_neighbor_distances(pts), 1e-7)
    scaling = np.repeat(np.log(np.sqrt(nn) * init_scaling)[:, None], 3, 1).astype(np.float32)
    rotation = np.zeros((pts.shape[0], 4), np.float32)
    rotation[:, 0] = 1
    opacity = np.full((pts.shape[0], 1), np.log(init_opacity / (1 - init_opacity)), np.float32)
    K = (sh_degree + 1) ** 2
    sh = np.zeros((pts.shape[0], K, 3), np.float32)
    sh[:, 0] = (cols - 0.5) / 0.28209479177387814
    t = lambda a: torch.from_numpy(a).to(device)  # noqa: E731
    return SplatData(means=t(pts), sh=t(sh), scaling_raw=t(scaling), rotation_raw=t(rotation), opacity_raw=t(opacity),
                     active_sh_degree=0), scene_scale
