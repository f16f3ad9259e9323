"""Deterministic COLMAP models (written with struct.pack / text formatting straight from COLMAP's published file layout, independent of
the product's reader) shared by tests/golden/gen_colmap_ref_golden.py and tests/test_io.py.  Every camera model the reference's reader
accepts (src/loader/formats/colmap.cpp:684-840) appears once; PNG files of the database's sizes are written beside the model (the
reader probes the first image, colmap.cpp:852-877)."""
import struct
import zlib

import numpy as np

# (camera_id, model_id, model name, width, height, params)
CAMERAS = [
    (1, 1, "PINHOLE", 640, 480, [500.0, 510.0, 320.0, 240.0]),
    (2, 0, "SIMPLE_PINHOLE", 800, 600, [700.0, 400.0, 300.0]),
    (3, 4, "OPENCV", 1000, 800, [900.0, 905.0, 500.0, 400.0, 0.1, -0.05, 0.001, 0.002]),
    (4, 5, "OPENCV_FISHEYE", 1000, 800, [400.0, 405.0, 500.0, 400.0, 0.01, 0.02, 0.03, 0.04]),
    (5, 2, "SIMPLE_RADIAL", 640, 480, [450.0, 320.0, 240.0, 0.0]),
    (6, 2, "SIMPLE_RADIAL", 640, 480, [450.0, 320.0, 240.0, 0.07]),
    (7, 3, "RADIAL", 640, 480, [455.0, 321.0, 239.0, 0.05, -0.01]),
    (8, 6, "FULL_OPENCV", 1200, 900, [1000.0, 1001.0, 600.0, 450.0, 0.11, -0.06, 0.0011, 0.0021, 0.013, 0.004, -0.003, 0.0007]),
    (9, 8, "SIMPLE_RADIAL_FISHEYE", 512, 512, [300.0, 256.0, 256.0, 0.02]),
    (10, 9, "RADIAL_FISHEYE", 512, 512, [301.0, 255.0, 257.0, 0.021, -0.004]),
]


def poses(n, seed=7):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        q = rng.standard_normal(4)
        q /= np.linalg.norm(q)
        out.append((q, rng.standard_normal(3) * 2.0, f"frame_{i:03d}.png"))
    return out


def points(n=11, seed=3):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((n, 3)) * 3.0, rng.integers(0, 256, (n, 3))


def write_png(path, w, h):
    raw = b"".join(b"\0" + bytes(3 * w) for _ in range(h))
    chunk = lambda t, d: struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))  # noqa: E731
    path.write_bytes(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 9)) +
                     chunk(b"IEND", b""))


def write_model(root, text=False, cameras=CAMERAS, images_folder="images", image_size=None, order=None):
    """`image_size`: (w, h) of the PNG files instead of each camera's database size divided by the folder factor (the dimension-correction
    case); `order`: permutation of the image records (the readers keep file order)."""
    sp = root / "sparse" / "0"
    sp.mkdir(parents=True)
    ps = poses(len(cameras))
    pts, cols = points()
    idx = list(range(len(cameras))) if order is None else list(order)
    if text:
        (sp / "cameras.txt").write_text("# Camera list with one line of data per camera:\n#   CAMERA_ID, MODEL, WIDTH, HEIGHT, PARAMS[]\n" + "".join(
            f"{cid} {name} {w} {h} " + " ".join(repr(p) for p in par) + "\n" for cid, _, name, w, h, par in cameras))
        (sp / "images.txt").write_text("# Image list with two lines of data per image:\n" + "".join(
            f"{100 + i} " + " ".join(repr(float(v)) for v in list(ps[i][0]) + list(ps[i][1])) + f" {cameras[i][0]} {ps[i][2]}\n" +
            ("12.5 7.25 -1 " * ((i + 1) % 3)).strip() + "\n" for i in idx))   # (empty observation lines in the middle of the file; the
        # reference's reader rejects a file whose LAST image has an empty one: colmap.cpp:481-484 + 516-519)
        (sp / "points3D.txt").write_text("# 3D point list with one line of data per point:\n" + "".join(
            f"{i} " + " ".join(repr(float(v)) for v in pts[i]) + " " + " ".join(str(int(c)) for c in cols[i]) + " 0.5" + " 1 2" * (i % 3) + "\n"
            for i in range(len(pts))))
    else:
        b = struct.pack("<Q", len(cameras))
        for cid, mid, _, w, h, par in cameras:
            b += struct.pack("<IiQQ", cid, mid, w, h) + struct.pack("<%dd" % len(par), *par)
        (sp / "cameras.bin").write_bytes(b)
        b = struct.pack("<Q", len(cameras))
        for i in idx:
            q, t, name = ps[i]
            b += struct.pack("<I4d3dI", 100 + i, *q, *t, cameras[i][0]) + name.encode() + b"\0" + struct.pack("<Q", i % 3) + b"\0" * (24 * (i % 3))
        (sp / "images.bin").write_bytes(b)
        b = struct.pack("<Q", len(pts))
        for i in range(len(pts)):
            b += struct.pack("<Q3d3BdQ", i, *pts[i], *[int(c) for c in cols[i]], 0.5, i % 3) + b"\0" * (8 * (i % 3))
        (sp / "points3D.bin").write_bytes(b)
    (root / images_folder).mkdir()
    suffix = images_folder.rsplit("_", 1)[-1] if "_" in images_folder else ""
    factor = int(suffix) if suffix.isdigit() else 1
    for i, (_, _, _, w, h, _) in enumerate(cameras):
        sz = image_size if image_size is not None else (w // factor, h // factor)
        write_png(root / images_folder / ps[i][2], *sz)
    return ps, pts, cols


# name -> keyword arguments of write_model + the images folder handed to the readers
CASES = {
    "bin": dict(text=False),
    "text": dict(text=True),
    "bin_images_2": dict(text=False, images_folder="images_2"),
    "text_images_4": dict(text=True, images_folder="images_4"),
    "bin_resized_files": dict(text=False, image_size=(960, 540)),      # first image is not the database's 640x480: every camera is corrected
    "bin_reordered": dict(text=False, order=[3, 0, 9, 1, 2, 8, 4, 7, 5, 6]),
}


def parse_tool_output(text):
    """colmap_ref_tool's stdout -> dict(center, cameras=[dict], points [N,3], colors [N,3])."""
    lines = text.strip().split("\n")
    out = {"cameras": []}
    k = 0
    while k < len(lines):
        ln = lines[k]
        if ln.startswith("center"):
            out["center"] = np.array(ln.split()[1:], np.float32)
        elif ln.startswith("camera "):
            head, name, R, T, rad, tan = [s.strip() for s in ln.split("|")]
            h = head.split()
            out["cameras"].append(dict(uid=int(h[1]), camera_id=int(h[2]), model=int(h[3]), model_type=int(h[4]), width=int(h[5]), height=int(h[6]),
                                       fx=np.float32(h[7]), fy=np.float32(h[8]), cx=np.float32(h[9]), cy=np.float32(h[10]), name=name,
                                       R=np.array(R.split(), np.float32).reshape(3, 3), T=np.array(T.split(), np.float32),
                                       radial=np.array(rad.split(), np.float32), tangential=np.array(tan.split(), np.float32)))
        elif ln.startswith("points "):
            n = int(ln.split()[1])
            a = np.array([r.split() for r in lines[k + 1:k + 1 + n]], np.float32).reshape(n, 6)
            out["points"], out["colors"] = a[:, :3], a[:, 3:]
            k += n
        elif ln.startswith("error"):
            out["error"] = ln[6:]
        k += 1
    return out
