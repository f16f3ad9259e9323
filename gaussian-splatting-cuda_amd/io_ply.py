"""3DGS PLY import / export: the model exchange format of the reference and of every 3DGS viewer.

Export follows SplatData::to_point_cloud / write_ply_impl (src/core/splat_data.cpp:113-169, 401-420, 484-502): one binary
little-endian `vertex` element of float properties, in this order:
    x y z  nx ny nz (zeros)  f_dc_0..2  f_rest_0..(3(K-1)-1)  opacity  scale_0..2  rot_0..3
with sh0 / shN stored CHANNEL-major (`transpose(1, 2).flatten(1)`: all red coefficients, then green, then blue), raw
(pre-activation) opacity and scales, and the quaternion normalised.  Import follows load_ply (src/loader/formats/ply.cpp:
445-476, 497-640): properties are located by name (any order, extra properties ignored), SH blocks converted back to
[N,B,3], missing blocks defaulted (opacity 0, log-scale, identity quaternion), SH degree = sqrt(B_rest + 1) - 1."""
import os

import numpy as np
import torch

from .rasterizer import SplatData

_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "<i2", "int16": "<i2", "ushort": "<u2", "uint16": "<u2",
              "int": "<i4", "int32": "<i4", "uint": "<u4", "uint32": "<u4", "float": "<f4", "float32": "<f4", "double": "<f8",
              "float64": "<f8"}
DEFAULT_LOG_SCALE = -5.0  # ply.cpp ply_constants


def attribute_names(K):
    """SplatData::get_attribute_names (splat_data.cpp:401-420)."""
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(3)]
    names += [f"f_rest_{i}" for i in range(3 * (K - 1))]
    names += ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)]
    return names


def save_ply(model: SplatData, path, iteration=None, stem=""):
    """`path` is a directory when `iteration` / `stem` is given (splat_<iteration>.ply or <stem>.ply inside it, as
    write_ply_impl names its files), else the file itself."""
    if iteration is not None or stem:
        os.makedirs(path, exist_ok=True)
        path = os.path.join(path, (stem + ".ply") if stem else f"splat_{iteration}.ply")
    with torch.no_grad():
        means = model.means.detach().float().cpu().numpy()
        sh = model.sh.detach().float().cpu().numpy()                       # [N,K,3]
        opac = model.opacity_raw.detach().float().cpu().numpy().reshape(-1, 1)
        scal = model.scaling_raw.detach().float().cpu().numpy()
        rot = torch.nn.functional.normalize(model.rotation_raw.detach().float(), dim=-1).cpu().numpy()
    N, K = sh.shape[0], sh.shape[1]
    cols = [means, np.zeros_like(means), sh[:, :1].transpose(0, 2, 1).reshape(N, -1), sh[:, 1:].transpose(0, 2, 1).reshape(N, -1), opac, scal, rot]
    data = np.ascontiguousarray(np.concatenate(cols, axis=1), dtype="<f4")
    names = attribute_names(K)
    assert data.shape[1] == len(names)
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % N
    header += "".join(f"property float {n}\n" for n in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        data.tofile(f)
    return path


def _parse_header(buf):
    end = buf.find(b"end_header")
    if end < 0 or not buf.startswith(b"ply"):
        raise RuntimeError("not a PLY file")
    nl = buf.find(b"\n", end)
    lines = buf[:end].decode("ascii", "replace").splitlines()
    fmt, count, props, in_vertex, seen_vertex = None, 0, [], False, False
    for ln in lines:
        tok = ln.split()
        if not tok:
            continue
        if tok[0] == "format":
            fmt = tok[1]
        elif tok[0] == "element":
            in_vertex = tok[1] == "vertex"
            if in_vertex:
                count, seen_vertex = int(tok[2]), True
            elif not seen_vertex:
                raise RuntimeError("PLY: elements before `vertex` are not supported")
        elif tok[0] == "property" and in_vertex:
            if tok[1] == "list":
                raise RuntimeError("PLY: list properties in the vertex element are not supported")
            if tok[1] not in _PLY_TYPES:
                raise RuntimeError(f"PLY: unknown property type {tok[1]}")
            props.append((tok[2], _PLY_TYPES[tok[1]]))
    if fmt != "binary_little_endian":
        raise RuntimeError(f"PLY: only binary_little_endian is supported (got {fmt})")
    if not seen_vertex:
        raise RuntimeError("PLY: no vertex element")
    return nl + 1, count, props


def load_ply(path, device="cpu"):
    if not os.path.exists(path):
        raise RuntimeError(f"PLY file does not exist: {path}")
    buf = np.memmap(path, dtype=np.uint8, mode="r")
    off, count, props = _parse_header(bytes(buf[:min(len(buf), 1 << 20)]))
    dt = np.dtype({"names": [p[0] for p in props], "formats": [p[1] for p in props]})
    if off + count * dt.itemsize > len(buf):
        raise RuntimeError("PLY: file truncated")
    v = np.frombuffer(buf, dtype=dt, count=count, offset=off)
    names = set(dt.names)

    def col(n, default=0.0):
        return v[n].astype(np.float32) if n in names else np.full(count, default, np.float32)

    means = np.stack([col("x"), col("y"), col("z")], 1)

    def sh_block(prefix, fallback_b):
        idx = sorted(int(n[len(prefix):]) for n in names if n.startswith(prefix) and n[len(prefix):].isdigit())
        if not idx or len(idx) % 3 != 0:
            return np.zeros((count, fallback_b, 3), np.float32)
        B = len(idx) // 3
        flat = np.stack([col(f"{prefix}{j}") for j in range(len(idx))], 1)        # j = channel * B + b
        return np.ascontiguousarray(flat.reshape(count, 3, B).transpose(0, 2, 1))   # [N,B,3]

    sh0, shN = sh_block("f_dc_", 1), sh_block("f_rest_", 15)
    opacity = col("opacity").reshape(-1, 1)
    has_scale = all(f"scale_{i}" in names for i in range(3))
    scaling = np.stack([col(f"scale_{i}") for i in range(3)], 1) if has_scale else np.full((count, 3), DEFAULT_LOG_SCALE, np.float32)
    has_rot = all(f"rot_{i}" in names for i in range(4))
    rotation = np.stack([col(f"rot_{i}") for i in range(4)], 1) if has_rot else np.tile(np.array([[1, 0, 0, 0]], np.float32), (count, 1))
    sh = np.concatenate([sh0, shN], 1)
    degree = int(np.sqrt(shN.shape[1] + 1)) - 1
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)  # noqa: E731
    return SplatData(means=t(means), sh=t(sh), scaling_raw=t(scaling), rotation_raw=t(rotation), opacity_raw=t(opacity),
                     active_sh_degree=degree)


def load_point_cloud_ply(path):
    """load_simple_ply_point_cloud (src/loader/formats/transforms.cpp:264-380): positions float32 [N,3] and colours uint8 [N,3]
    (red / green / blue properties, white when absent) of a plain point-cloud PLY, e.g. the `points3d.ply` of a Blender scene."""
    if not os.path.exists(path):
        raise RuntimeError(f"PLY file not found: {path}")
    buf = np.memmap(path, dtype=np.uint8, mode="r")
    off, count, props = _parse_header(bytes(buf[:min(len(buf), 1 << 20)]))
    dt = np.dtype({"names": [p[0] for p in props], "formats": [p[1] for p in props]})
    if off + count * dt.itemsize > len(buf):
        raise RuntimeError("PLY: file truncated")
    v = np.frombuffer(buf, dtype=dt, count=count, offset=off)
    if not all(n in dt.names for n in ("x", "y", "z")):
        raise RuntimeError("PLY file missing vertex positions")
    xyz = np.stack([v["x"], v["y"], v["z"]], 1).astype(np.float32)
    if all(n in dt.names for n in ("red", "green", "blue")):
        rgb = np.stack([v["red"], v["green"], v["blue"]], 1)
        rgb = rgb.astype(np.uint8) if rgb.dtype.kind in "ui" else np.clip(rgb * 255.0, 0, 255).astype(np.uint8)
    else:
        rgb = np.full((count, 3), 255, np.uint8)
    return np.ascontiguousarray(xyz), np.ascontiguousarray(rgb)
