"""Model / dataset IO: 3DGS PLY layout and round trip, COLMAP binary sparse model, point-cloud initialisation."""
import os
import struct

import numpy as np
import pytest
import torch


def _model(N=37, K=16, seed=0):
    import gsx  # noqa: F401
    from gsx.rasterizer import SplatData
    g = torch.Generator().manual_seed(seed)
    return SplatData(means=torch.randn(N, 3, generator=g), sh=torch.randn(N, K, 3, generator=g), scaling_raw=torch.randn(N, 3, generator=g),
                     rotation_raw=torch.randn(N, 4, generator=g), opacity_raw=torch.randn(N, 1, generator=g), active_sh_degree=3)


@pytest.mark.parametrize("K", [16, 4, 1])
def test_ply_layout_and_round_trip(tmp_path, K):
    from gsx import io_ply
    m = _model(K=K)
    path = io_ply.save_ply(m, str(tmp_path), iteration=7000)
    assert path.endswith("splat_7000.ply")
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    lines = head.decode().splitlines()
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 37"]
    names = [ln.split()[2] for ln in lines[3:]]
    assert names == io_ply.attribute_names(K) and all(ln.startswith("property float ") for ln in lines[3:])
    assert names[:9] == ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] and names[-8:] == ["opacity", "scale_0", "scale_1",
                                                                                                        "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    rows = np.frombuffer(body, "<f4").reshape(37, len(names))
    assert np.array_equal(rows[:, 3:6], np.zeros((37, 3), np.float32))                 # normals
    assert np.array_equal(rows[:, 6:9], m.sh[:, 0].numpy())                            # f_dc = (r, g, b) of basis 0
    if K > 1:  # f_rest is channel-major: all red coefficients first (transpose(1,2).flatten(1), splat_data.cpp:492-493)
        assert np.array_equal(rows[:, 9:9 + (K - 1)], m.sh[:, 1:, 0].numpy())
        assert np.array_equal(rows[:, 9 + 2 * (K - 1):9 + 3 * (K - 1)], m.sh[:, 1:, 2].numpy())
    q = rows[:, -4:]
    np.testing.assert_allclose(np.linalg.norm(q, axis=1), 1.0, rtol=1e-6)              # quaternion stored normalised
    back = io_ply.load_ply(path)
    assert torch.equal(back.means, m.means) and torch.equal(back.opacity_raw, m.opacity_raw) and torch.equal(back.scaling_raw, m.scaling_raw)
    if K > 1:
        assert torch.equal(back.sh, m.sh) and back.active_sh_degree == int(np.sqrt(K)) - 1
    else:  # no f_rest block: the loader substitutes 15 zero bands (ply.cpp:552-554)
        assert torch.equal(back.sh[:, :1], m.sh) and back.sh.shape[1] == 16 and float(back.sh[:, 1:].abs().max()) == 0.0
    assert torch.allclose(back.rotation_raw, torch.nn.functional.normalize(m.rotation_raw, dim=-1), atol=1e-7)


def test_ply_loader_locates_properties_by_name(tmp_path):
    from gsx import io_ply
    # shuffled order, a double and an uchar property in between, no rotation / f_rest
    dt = np.dtype([("opacity", "<f4"), ("junk", "u1"), ("z", "<f4"), ("x", "<f8"), ("y", "<f4"), ("f_dc_2", "<f4"), ("f_dc_0", "<f4"),
                   ("f_dc_1", "<f4"), ("scale_0", "<f4"), ("scale_1", "<f4"), ("scale_2", "<f4")])
    v = np.zeros(5, dt)
    for n in dt.names:
        v[n] = np.arange(5) + (hash(n) % 7)
    types = {"<f4": "float", "<f8": "double", "u1": "uchar", "|u1": "uchar"}
    header = "ply\nformat binary_little_endian 1.0\ncomment made by hand\nelement vertex 5\n" + "".join(
        f"property {types[dt[n].str]} {n}\n" for n in dt.names) + "end_header\n"
    p = tmp_path / "x.ply"
    p.write_bytes(header.encode() + v.tobytes())
    m = io_ply.load_ply(str(p))
    assert np.array_equal(m.means.numpy(), np.stack([v["x"], v["y"], v["z"]], 1).astype(np.float32))
    assert np.array_equal(m.sh[:, 0].numpy(), np.stack([v["f_dc_0"], v["f_dc_1"], v["f_dc_2"]], 1))
    assert np.array_equal(m.rotation_raw.numpy(), np.tile([[1, 0, 0, 0]], (5, 1)).astype(np.float32))    # identity default
    assert m.sh.shape == (5, 16, 3)
    with pytest.raises(RuntimeError):
        io_ply.load_ply(str(tmp_path / "missing.ply"))
    (tmp_path / "ascii.ply").write_bytes(b"ply\nformat ascii 1.0\nelement vertex 0\nend_header\n")
    with pytest.raises(RuntimeError):
        io_ply.load_ply(str(tmp_path / "ascii.ply"))


def _write_colmap(root, trailing=False):
    sp = root / "sparse" / "0"
    sp.mkdir(parents=True)
    cams = [(1, 1, 640, 480, [500.0, 510.0, 320.0, 240.0]),                               # PINHOLE
            (2, 0, 800, 600, [700.0, 400.0, 300.0]),                                      # SIMPLE_PINHOLE
            (3, 4, 1000, 800, [900.0, 905.0, 500.0, 400.0, 0.1, -0.05, 0.001, 0.002]),    # OPENCV
            (4, 5, 1000, 800, [400.0, 405.0, 500.0, 400.0, 0.01, 0.02, 0.03, 0.04])]      # OPENCV_FISHEYE
    b = struct.pack("<Q", len(cams))
    for cid, mid, w, h, ps in cams:
        b += struct.pack("<IiQQ", cid, mid, w, h) + struct.pack("<%dd" % len(ps), *ps)
    (sp / "cameras.bin").write_bytes(b + (b"x" if trailing else b""))
    rng = np.random.default_rng(0)
    imgs = []
    b = struct.pack("<Q", 4)
    for i in range(4):
        q = rng.standard_normal(4)
        q /= np.linalg.norm(q)
        t = rng.standard_normal(3)
        name = f"img_{i}.png"
        npts = i  # a few fake 2-D observations to skip over
        b += struct.pack("<I4d3dI", 10 + i, *q, *t, i + 1) + name.encode() + b"\0" + struct.pack("<Q", npts) + b"\0" * (24 * npts)
        imgs.append((q, t, name))
    (sp / "images.bin").write_bytes(b)
    pts = rng.standard_normal((6, 3))
    cols = rng.integers(0, 255, (6, 3))
    b = struct.pack("<Q", 6)
    for i in range(6):
        b += struct.pack("<Q3d3BdQ", i, *pts[i], *[int(c) for c in cols[i]], 0.5, i % 3) + b"\0" * (8 * (i % 3))
    (sp / "points3D.bin").write_bytes(b)
    (root / "images").mkdir()
    return cams, imgs, pts, cols


def test_colmap_binary_reader(tmp_path):
    import gsx  # noqa: F401
    from gsx import io_colmap, ops
    cams, imgs, pts, cols = _write_colmap(tmp_path)
    sc = io_colmap.load_colmap(str(tmp_path))
    assert len(sc.cameras) == 4 and [c.image_name for c in sc.cameras] == [f"img_{i}.png" for i in range(4)]
    for i, c in enumerate(sc.cameras):
        q, t, _ = imgs[i]
        R = io_colmap.qvec2rotmat(q.astype(np.float32))
        vm = c.camera.viewmat.numpy()
        np.testing.assert_allclose(vm[:3, :3], R, atol=1e-6)
        np.testing.assert_allclose(vm[:3, 3], t.astype(np.float32), atol=1e-6)             # world -> camera [R|t]
        np.testing.assert_allclose(R @ sc.camera_locations[i] + t, 0, atol=1e-5)           # centre = -R^T t
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-5)
    K0, K1 = sc.cameras[0].camera.K.numpy(), sc.cameras[1].camera.K.numpy()
    assert (K0[0, 0], K0[1, 1], K0[0, 2], K0[1, 2]) == (500.0, 510.0, 320.0, 240.0)
    assert (K1[0, 0], K1[1, 1], K1[0, 2], K1[1, 2]) == (700.0, 700.0, 400.0, 300.0)        # SIMPLE_PINHOLE: one focal
    c2, c3 = sc.cameras[2].camera, sc.cameras[3].camera
    assert c2.camera_model == ops.CameraModelType.PINHOLE and c3.camera_model == ops.CameraModelType.FISHEYE
    np.testing.assert_allclose(c2.radial.numpy(), [0.1, -0.05, 0, 0], atol=1e-7)           # padded to 4 (rasterizer.cpp:183-195)
    np.testing.assert_allclose(c2.tangential.numpy(), [0.001, 0.002, 0, 0], atol=1e-7)
    np.testing.assert_allclose(c3.radial.numpy(), [0.01, 0.02, 0.03, 0.04], atol=1e-7)
    assert (c2.width, c2.height) == (1000, 800)
    np.testing.assert_allclose(sc.points, pts.astype(np.float32))
    assert np.array_equal(sc.colors, cols.astype(np.uint8))
    # images_2: intrinsics and size divided by the folder's factor (colmap.cpp:172-258)
    (tmp_path / "images_2").mkdir()
    half = io_colmap.load_colmap(str(tmp_path), images_folder="images_2")
    Kh = half.cameras[0].camera.K.numpy()
    assert (Kh[0, 0], Kh[1, 1], Kh[0, 2], Kh[1, 2]) == (250.0, 255.0, 160.0, 120.0) and half.cameras[0].camera.width == 320


def test_colmap_trailing_bytes_rejected(tmp_path):
    import gsx  # noqa: F401
    from gsx import io_colmap
    _write_colmap(tmp_path, trailing=True)
    with pytest.raises(RuntimeError, match="trailing"):
        io_colmap.load_colmap(str(tmp_path))


def test_init_model_from_pointcloud():
    import gsx  # noqa: F401
    from gsx import io_colmap
    pts = np.array([[0, 0, 0], [1, 0, 0], [0, 2, 0], [0, 0, 3], [0, 0, 0]], np.float32)     # a duplicate point (distance 0 is skipped)
    cols = np.array([[255, 0, 128]] * 5, np.uint8)
    m, scale = io_colmap.init_model_from_pointcloud(pts, cols, np.zeros(3), sh_degree=3)
    assert m.sh.shape == (5, 16, 3) and m.active_sh_degree == 0
    np.testing.assert_allclose(m.sh[0, 0].numpy(), (np.array([1, 0, 128 / 255]) - 0.5) / 0.28209479177387814, rtol=1e-5, atol=2e-7)
    assert float(m.sh[:, 1:].abs().max()) == 0 and torch.equal(m.rotation_raw[:, 0], torch.ones(5))
    np.testing.assert_allclose(torch.sigmoid(m.opacity_raw).numpy(), 0.5, rtol=1e-6)
    # point 0: the 4 nearest are itself (0), its duplicate (0), then 1, 2 -> valid neighbours {1, 2} (3 is not among the 4 results)
    np.testing.assert_allclose(float(m.scaling_raw[0, 0]), np.log(np.sqrt(1.5) * 0.1), rtol=1e-5)
    assert scale == float(np.median(np.linalg.norm(pts, axis=1)))


def test_image_io_sizes_and_round_trip(tmp_path):
    import gsx  # noqa: F401
    from gsx import io_image
    from PIL import Image
    rng = np.random.default_rng(0)
    rgba = rng.integers(0, 255, (48, 80, 4), dtype=np.uint8)
    Image.fromarray(rgba, "RGBA").save(tmp_path / "a.png")
    a = io_image.load_image(str(tmp_path / "a.png"))
    assert a.shape == (48, 80, 3) and np.array_equal(a, rgba[:, :, :3])                     # alpha dropped, bit-identical
    assert io_image.load_image(str(tmp_path / "a.png"), res_div=2).shape == (24, 40, 3)
    assert io_image.load_image(str(tmp_path / "a.png"), res_div=4, max_width=16).shape == (9, 16, 3)   # 16 * 12 // 20 = 9
    assert io_image.target_size(100, 300, 1, 150) == (50, 150) and io_image.target_size(7, 7, 8, 0) == (1, 1)
    half = io_image.load_image(str(tmp_path / "a.png"), res_div=2).astype(np.float32)
    box = rgba[:, :, :3].astype(np.float32).reshape(24, 2, 40, 2, 3).mean((1, 3))
    assert np.abs(half - box).max() <= 1.0                                                   # box reduction for exact divisors
    Image.fromarray(rgba[:, :, 0], "L").save(tmp_path / "g.png")
    g = io_image.load_image(str(tmp_path / "g.png"))
    assert g.shape == (48, 80, 3) and np.array_equal(g[:, :, 0], g[:, :, 2])                 # grey -> RGB
    t = io_image.load_and_get_image(str(tmp_path / "a.png"))
    assert t.shape == (3, 48, 80) and t.dtype == torch.float32 and float(t.max()) <= 1.0
    io_image.save_image(str(tmp_path / "b.png"), t)
    assert np.array_equal(io_image.load_image(str(tmp_path / "b.png")), rgba[:, :, :3])      # float -> 8 bit -> same bytes
    with pytest.raises(ValueError):
        io_image.load_image(str(tmp_path / "a.png"), res_div=3)
    with pytest.raises(RuntimeError):
        io_image.load_image(str(tmp_path / "missing.png"))


def test_resample_restates_oiio_resample_interpolate():
    """io_image.resample_oiio = OIIO::ImageBufAlgo::resample(dst, src, interpolate = true) as the reference calls it
    (src/core/image_io.cpp:33-49).  OpenImageIO is not in this image (vcpkg dependency): pinned by fixtures computed by hand from its
    published algorithm — sample the source at the destination pixel's CENTRE, bilinear between the four texel centres, texels as
    v / 255, store (uint8)(f * 255 + 0.5)."""
    import gsx  # noqa: F401
    from gsx import io_image
    # (1) exact halving: the centre of a destination pixel sits on the corner shared by a 2 x 2 source block: weights 1/4 each
    src = np.array([[[10], [20], [7], [9]], [[30], [41], [1], [2]]], np.uint8)              # [H=2, W=4, C=1]
    out = io_image.resample_oiio(src, 2, 1)
    # (10 + 20 + 30 + 41) / 4 = 25.25 -> +0.5 -> 25 ; (7 + 9 + 1 + 2) / 4 = 4.75 -> +0.5 -> 5
    assert out.shape == (1, 2, 1) and out[0, :, 0].tolist() == [25, 5]
    # (2) 5 -> 3 along x, one row: centres at 5/6, 15/6, 25/6 -> minus 0.5 -> 1/3 (texel 0, frac 1/3), 2 (texel 2, frac 0), 11/3 (texel 3, frac 2/3)
    row = np.array([[[0], [90], [77], [30], [120]]], np.uint8)
    out = io_image.resample_oiio(row, 3, 1)
    # y: one source row -> centre 0.5 -> minus 0.5 -> texel 0, frac 0 (the row below is outside: weight 0)
    # 2/3 * 0 + 1/3 * 90 = 30 ; 77 ; 1/3 * 30 + 2/3 * 120 = 90
    assert out[0, :, 0].tolist() == [30, 77, 90]
    # (3) 3 -> 2 along y, three channels: centres at 0.75, 2.25 -> 0.25 (texel 0, frac 1/4), 1.75 (texel 1, frac 3/4)
    col = np.array([[[0, 255, 8]], [[100, 155, 8]], [[200, 55, 9]]], np.uint8)
    out = io_image.resample_oiio(col, 1, 2)
    # 0.75 * 0 + 0.25 * 100 = 25 ; 0.75 * 255 + 0.25 * 155 = 230 ; 8        |  0.25 * 100 + 0.75 * 200 = 175 ; 0.25 * 155 + 0.75 * 55 = 80 ; 8.75 -> 9
    assert out[:, 0, :].tolist() == [[25, 230, 8], [175, 80, 9]]
    # (4) identity size: every pixel samples its own centre
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, (7, 5, 3), dtype=np.uint8)
    assert np.array_equal(io_image.resample_oiio(a, 5, 7), a)


def test_colmap_text_reader_matches_binary(tmp_path):
    import gsx  # noqa: F401
    from gsx import io_colmap
    root_b, root_t = tmp_path / "bin", tmp_path / "txt"
    root_b.mkdir()
    cams, imgs, pts, cols = _write_colmap(root_b)
    sp = root_t / "sparse" / "0"
    sp.mkdir(parents=True)
    (root_t / "images").mkdir()
    names = {0: "SIMPLE_PINHOLE", 1: "PINHOLE", 4: "OPENCV", 5: "OPENCV_FISHEYE"}
    (sp / "cameras.txt").write_text("# Camera list\n" + "".join(
        f"{cid} {names[mid]} {w} {h} " + " ".join(repr(p) for p in ps) + "\n" for cid, mid, w, h, ps in cams))
    (sp / "images.txt").write_text("# Image list\n" + "".join(
        f"{10 + i} " + " ".join(repr(float(v)) for v in list(q) + list(t)) + f" {i + 1} {name}\n" + ("1.0 2.0 -1 " * i).strip() + "\n"
        for i, (q, t, name) in enumerate(imgs)))
    (sp / "points3D.txt").write_text("# 3D point list\n" + "".join(
        f"{i} " + " ".join(repr(float(v)) for v in pts[i]) + " " + " ".join(str(int(c)) for c in cols[i]) + " 0.5 1 2\n" for i in range(6)))
    a, b = io_colmap.load_colmap(str(root_b)), io_colmap.load_colmap(str(root_t))
    assert len(a.cameras) == len(b.cameras) == 4
    for ca, cb in zip(a.cameras, b.cameras):
        assert ca.image_name == cb.image_name and ca.model == cb.model
        assert torch.equal(ca.camera.viewmat, cb.camera.viewmat) and torch.equal(ca.camera.K, cb.camera.K)
    assert np.array_equal(a.points, b.points) and np.array_equal(a.colors, b.colors)
    with pytest.raises(RuntimeError):
        io_colmap.load_colmap(str(tmp_path / "nowhere"))


def test_transforms_json_reader(tmp_path):
    import json
    import gsx  # noqa: F401
    from gsx import io_transforms
    rng = np.random.default_rng(4)
    frames = []
    for i in range(3):
        q = rng.standard_normal(4)
        q /= np.linalg.norm(q)
        from gsx.io_colmap import qvec2rotmat
        c2w = np.eye(4)
        c2w[:3, :3] = qvec2rotmat(q.astype(np.float32))
        c2w[:3, 3] = rng.standard_normal(3)
        frames.append({"file_path": f"./train/r_{i}", "transform_matrix": c2w.tolist()})
    (tmp_path / "transforms_train.json").write_text(json.dumps({"camera_angle_x": 0.6911, "w": 800, "h": 800, "frames": frames}))
    sc = io_transforms.load_transforms(str(tmp_path))
    assert len(sc.cameras) == 3 and sc.cameras[0].image_name == "r_0"
    K = sc.cameras[0].camera.K.numpy()
    f = 0.5 * 800 / np.tan(0.5 * 0.6911)
    assert abs(K[0, 0] - f) < 1e-3 and K[1, 1] == K[0, 0] and (K[0, 2], K[1, 2]) == (400.0, 400.0)
    for fr, c in zip(frames, sc.cameras):
        c2w = np.array(fr["transform_matrix"])
        vm = c.camera.viewmat.numpy().astype(np.float64)
        # the reference also turns the WORLD by pi about y (w2c * RotY(pi), transforms.cpp:209-210): centre and viewing direction
        # (the camera looks along -z of the Blender camera frame = +z of ours) come out with x and z negated
        flip = np.array([-1.0, 1.0, -1.0])
        np.testing.assert_allclose(-vm[:3, :3].T @ vm[:3, 3], flip * c2w[:3, 3], atol=1e-5)
        np.testing.assert_allclose(vm[2, :3], flip * -c2w[:3, 2], atol=1e-5)
        assert abs(np.linalg.det(vm[:3, :3]) - 1.0) < 1e-5
    (tmp_path / "bad.json").write_text(json.dumps({"camera_angle_x": 0.7, "w": 800, "h": 600, "frames": frames}))
    with pytest.raises(RuntimeError, match="camera_angle_y"):
        io_transforms.load_transforms(str(tmp_path / "bad.json"))
    (tmp_path / "dist.json").write_text(json.dumps({"fl_x": 500, "fl_y": 500, "w": 8, "h": 8, "k1": 0.1, "frames": frames}))
    with pytest.raises(RuntimeError, match="distortion"):
        io_transforms.load_transforms(str(tmp_path / "dist.json"))
    pts, cols = io_transforms.generate_random_point_cloud()
    assert pts.shape == (10000, 3) and cols.dtype == np.uint8 and np.abs(pts).max() <= 1.0
    torch.manual_seed(8128)
    assert np.array_equal(pts, (torch.rand(10000, 3) * 2.0 - 1.0).numpy())              # the reference's draw (torch seed 8128)

    # a frame without transform_matrix (or with a malformed one) is an error, as upstream (transforms.cpp:187-195): never skipped
    for mutate in (lambda fr: fr.pop("transform_matrix"), lambda fr: fr.__setitem__("transform_matrix", [[1, 0, 0, 0]] * 3)):
        bad = json.loads(json.dumps(frames))
        mutate(bad[1])
        (tmp_path / "transforms_train.json").write_text(json.dumps({"camera_angle_x": 0.6911, "w": 800, "h": 800, "frames": bad}))
        with pytest.raises(RuntimeError):
            io_transforms.load_transforms(str(tmp_path))


def test_point_cloud_ply_reader(tmp_path):
    import gsx  # noqa: F401
    from gsx import io_ply
    dt = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    v = np.zeros(7, dt)
    rng = np.random.default_rng(1)
    for n in ("x", "y", "z"):
        v[n] = rng.standard_normal(7)
    for n in ("red", "green", "blue"):
        v[n] = rng.integers(0, 255, 7)
    header = "ply\nformat binary_little_endian 1.0\nelement vertex 7\n" + "".join(
        f"property {'float' if dt[n].kind == 'f' else 'uchar'} {n}\n" for n in dt.names) + "end_header\n"
    (tmp_path / "pc.ply").write_bytes(header.encode() + v.tobytes())
    xyz, rgb = io_ply.load_point_cloud_ply(str(tmp_path / "pc.ply"))
    assert np.array_equal(xyz, np.stack([v["x"], v["y"], v["z"]], 1)) and np.array_equal(rgb, np.stack([v["red"], v["green"], v["blue"]], 1))
    dt2 = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4")])
    header2 = "ply\nformat binary_little_endian 1.0\nelement vertex 2\nproperty float x\nproperty float y\nproperty float z\nend_header\n"
    (tmp_path / "nocol.ply").write_bytes(header2.encode() + np.zeros(2, dt2).tobytes())
    xyz2, rgb2 = io_ply.load_point_cloud_ply(str(tmp_path / "nocol.ply"))
    assert xyz2.shape == (2, 3) and np.all(rgb2 == 255)


# ---- PLY interchange with the reference's own PLY library --------------------------------------------------------------------------
# oracle/_ref/ply_ref_tool = the reference's vendored tinyply (include/external/tinyply.hpp + src/core/tinyply.cpp, compiled where they
# lie by oracle/build_ref_ply.sh) behind a small driver: `write` issues exactly the calls of the reference's exporter
# (src/core/splat_data.cpp:119-162), `read` parses a file with the same library.
_PLY_TOOL = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "ply_ref_tool")
_needs_tool = pytest.mark.skipif(not os.path.exists(_PLY_TOOL), reason="oracle/_ref/ply_ref_tool not built (needs /root/reference: oracle/build_ref_ply.sh)")


def _reference_attribute_names(K):
    """splat_data.cpp:401-419 (get_attribute_names), restated independently of gsx.io_ply."""
    a = ["x", "y", "z", "nx", "ny", "nz"] + ["f_dc_%d" % i for i in range(3)] + ["f_rest_%d" % i for i in range(3 * (K - 1))] + ["opacity"]
    return a + ["scale_%d" % i for i in range(3)] + ["rot_%d" % i for i in range(4)]


@_needs_tool
@pytest.mark.parametrize("K", [16, 4])
def test_ply_written_by_the_reference_library_is_loaded(tmp_path, K):
    """A splat PLY produced by the reference's exporter calls on the reference's PLY library, from blocks laid out as
    SplatData::to_point_cloud does (splat_data.cpp:484-505: sh0 / shN as transpose(1,2).flatten(1), rotation normalised), loads into the
    tensors it was made from."""
    import subprocess
    from gsx import io_ply
    m = _model(N=53, K=K, seed=3)
    sh = m.sh.numpy()
    rot = torch.nn.functional.normalize(m.rotation_raw, dim=-1).numpy()
    blocks = [m.means.numpy(), np.zeros((53, 3), np.float32), sh[:, :1].transpose(0, 2, 1).reshape(53, -1), sh[:, 1:].transpose(0, 2, 1).reshape(53, -1),
              m.opacity_raw.numpy(), m.scaling_raw.numpy(), rot]
    names = _reference_attribute_names(K)
    buf, off = struct.pack("<ii", 53, len(blocks)), 0
    for b in blocks:
        b = np.ascontiguousarray(b, np.float32)
        joined = "\n".join(names[off:off + b.shape[1]]).encode()
        buf += struct.pack("<iii", b.shape[1], b.shape[1], len(joined)) + joined + b.tobytes()
        off += b.shape[1]
    assert off == len(names)
    (tmp_path / "blocks.bin").write_bytes(buf)
    r = subprocess.run([_PLY_TOOL, "write", str(tmp_path / "blocks.bin"), str(tmp_path / "ref.ply")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    back = io_ply.load_ply(str(tmp_path / "ref.ply"))
    assert torch.equal(back.means, m.means) and torch.equal(back.sh, m.sh) and torch.equal(back.opacity_raw, m.opacity_raw)
    assert torch.equal(back.scaling_raw, m.scaling_raw) and np.array_equal(back.rotation_raw.numpy(), rot)
    assert back.active_sh_degree == int(np.sqrt(K)) - 1
    # and byte for byte what our exporter writes for the same model (header included)
    ours = io_ply.save_ply(m, str(tmp_path), iteration=1)
    assert open(ours, "rb").read() == (tmp_path / "ref.ply").read_bytes()


@_needs_tool
def test_ply_written_by_us_is_parsed_by_the_reference_library(tmp_path):
    import subprocess
    from gsx import io_ply
    m = _model(N=41, K=16, seed=8)
    path = io_ply.save_ply(m, str(tmp_path), iteration=30000)
    r = subprocess.run([_PLY_TOOL, "read", path, str(tmp_path / "cols.bin")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    raw = (tmp_path / "cols.bin").read_bytes()
    rows, nprops, ln = struct.unpack_from("<iii", raw, 0)
    names = raw[12:12 + ln].decode().split("\n")
    cols = np.frombuffer(raw, "<f4", offset=12 + ln).reshape(nprops, rows)
    assert rows == 41 and names == _reference_attribute_names(16)
    col = {n: cols[i] for i, n in enumerate(names)}
    assert np.array_equal(np.stack([col["x"], col["y"], col["z"]], 1), m.means.numpy())
    assert np.array_equal(col["f_dc_1"], m.sh[:, 0, 1].numpy())
    assert np.array_equal(col["f_rest_0"], m.sh[:, 1, 0].numpy()) and np.array_equal(col["f_rest_15"], m.sh[:, 1, 1].numpy())   # channel-major rest block
    assert np.array_equal(col["f_rest_44"], m.sh[:, 15, 2].numpy())
    assert np.array_equal(col["opacity"], m.opacity_raw[:, 0].numpy()) and np.array_equal(col["scale_2"], m.scaling_raw[:, 2].numpy())
    np.testing.assert_allclose(np.stack([col["rot_%d" % i] for i in range(4)], 1), torch.nn.functional.normalize(m.rotation_raw, dim=-1).numpy(), atol=1e-7)


def _colmap_reference_outputs(name, kw, root):
    """What the reference's own reader parsed from case `name`: the committed golden (tests/golden/colmap_ref/, generated by
    gen_colmap_ref_golden.py) and, where oracle/_ref/colmap_ref_tool is built, a live run on the model just written under `root`."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import colmap_cases
    outs = [colmap_cases.parse_tool_output(open(os.path.join(os.path.dirname(__file__), "golden", "colmap_ref", name + ".txt")).read())]
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "colmap_ref_tool")
    if os.path.exists(tool):
        r = subprocess.run([tool, str(root), kw.get("images_folder", "images"), "text" if kw.get("text") else "bin"], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(colmap_cases.parse_tool_output(r.stdout))
    return outs


@pytest.mark.parametrize("name", ["bin", "text", "bin_images_2", "text_images_4", "bin_resized_files", "bin_reordered"])
def test_colmap_loader_agrees_with_the_reference_reader(tmp_path, name):
    """io_colmap.load_colmap against the reference's reader (src/loader/formats/colmap.cpp, compiled unmodified: oracle/build_ref_colmap.sh)
    on models this file's own struct.pack writer produces from COLMAP's published layout: every accepted camera model, binary and text,
    down-scaled image folders, the first-image dimension correction, image order.  Intrinsics, sizes, names, model mapping, distortion
    vectors, R, T and the point cloud must be EQUAL (fp32, bit for bit); the scene centre (a mean of -R^T t over the cameras) to 5e-7."""
    import sys
    import gsx  # noqa: F401
    from gsx import io_colmap, ops
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import colmap_cases
    kw = colmap_cases.CASES[name]
    colmap_cases.write_model(tmp_path, **kw)
    sc = io_colmap.load_colmap(str(tmp_path), images_folder=kw.get("images_folder", "images"))
    model_ids = {n: i for i, n in enumerate(["SIMPLE_PINHOLE", "PINHOLE", "SIMPLE_RADIAL", "RADIAL", "OPENCV", "OPENCV_FISHEYE", "FULL_OPENCV", "FOV",
                                             "SIMPLE_RADIAL_FISHEYE", "RADIAL_FISHEYE", "THIN_PRISM_FISHEYE"])}
    for ref in _colmap_reference_outputs(name, kw, tmp_path):
        assert "error" not in ref and len(ref["cameras"]) == len(sc.cameras) == len(colmap_cases.CAMERAS)
        for ours, rc in zip(sc.cameras, ref["cameras"]):
            cam = ours.camera
            assert ours.image_name == rc["name"] and ours.uid == rc["uid"] and model_ids[ours.model] == rc["model"]
            assert (cam.width, cam.height) == (rc["width"], rc["height"])
            assert int(cam.camera_model) == rc["model_type"]                       # gsplat::CameraModelType (PINHOLE 0, FISHEYE 2)
            K = cam.K.numpy()
            assert (K[0, 0], K[1, 1], K[0, 2], K[1, 2]) == (rc["fx"], rc["fy"], rc["cx"], rc["cy"]), (ours.image_name, K, rc)
            vm = cam.viewmat.numpy()
            assert np.array_equal(vm[:3, :3], rc["R"])
            assert np.array_equal(vm[:3, 3], rc["T"])
            for mine, theirs in ((cam.radial, rc["radial"]), (cam.tangential, rc["tangential"])):
                if theirs.size == 0:
                    assert mine is None
                else:   # ours is padded with zeros to >= 4 entries, as the reference pads at render time (rasterizer.cpp:183-195)
                    m = mine.numpy()
                    assert np.array_equal(m[:theirs.size], theirs) and not m[theirs.size:].any()
        np.testing.assert_allclose(sc.scene_center, ref["center"], rtol=0, atol=5e-7)
        assert np.array_equal(sc.points, ref["points"]) and np.array_equal(sc.colors.astype(np.float32), ref["colors"])
