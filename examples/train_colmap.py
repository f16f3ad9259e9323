"""BASELINE configs[2] as a command: train a COLMAP capture with the `--gut` path — what the reference's timing script runs per scene
(eval/timing_mipnerf360_default.sh:30-37: `LichtFeld-Studio -d data/garden --images images_4 --iter 30000 --headless --config
eval/default_optimization_params.json`), on this backend:

    python examples/train_colmap.py -d data/garden --images images_4 --iter 30000 [--config params.json] [--eval] [--test-every 8]
                                    [-o out_dir] [--json out.json] [--max-width 3840] [--resize-factor -1]

    COLMAP sparse model (io_colmap.load_colmap: cameras / images / points3D .bin or .txt, `images_<k>` scales the intrinsics as upstream)
    -> images decoded and resized as Camera::load_and_get_image does (io_image) -> SplatData::init_model_from_pointcloud
    -> trainer.Trainer (rasterize_fused -> fused L1 + SSIM loss -> backward -> MCMC strategy -> fused Adam), parameter preset "default" or --config
    -> one JSON line: iterations/s (whole loop incl. densification), seconds, Gaussian count, PSNR / SSIM over the test split (--eval: every
       `test_every`-th image is held out, src/training/dataset.hpp:42) or over every 8th training image; -o writes the model as a PLY.

The Mip-NeRF360 captures are not in this container (no network): `--make-synthetic DIR` writes a small capture of the same layout — binary
COLMAP sparse model + PNG images rendered from a hidden scene — so that the whole path (reader, image IO, initialisation, training, export)
runs end to end on files: `python examples/train_colmap.py --make-synthetic /tmp/cap && python examples/train_colmap.py -d /tmp/cap --iter 600 --eval`
(tests/test_gpu_training.py does exactly that).  Distorted pinhole and fisheye captures train through the same command (the camera model of each
image comes from the sparse model)."""
import argparse
import json
import math
import os
import struct
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import gsx  # noqa: E402,F401
from gsx import io_colmap, io_image, io_ply, metrics, parameters, rasterizer, scenes, trainer  # noqa: E402


def rotmat2qvec(R):
    """COLMAP's quaternion (w, x, y, z) of a rotation matrix (inverse of io_colmap.qvec2rotmat)."""
    K = np.array([[R[0, 0] - R[1, 1] - R[2, 2], 0, 0, 0], [R[1, 0] + R[0, 1], R[1, 1] - R[0, 0] - R[2, 2], 0, 0],
                  [R[2, 0] + R[0, 2], R[2, 1] + R[1, 2], R[2, 2] - R[0, 0] - R[1, 1], 0],
                  [R[1, 2] - R[2, 1], R[2, 0] - R[0, 2], R[0, 1] - R[1, 0], R[0, 0] + R[1, 1] + R[2, 2]]]) / 3.0
    w, v = np.linalg.eigh(K)
    q = v[[3, 0, 1, 2], np.argmax(w)]
    return -q if q[0] < 0 else q


def make_synthetic(root, n_cams=24, width=320, height=208, n_gt=20_000, n_init=6_000, dev="cuda:0", seed=5):
    """Writes a capture: <root>/sparse/0/{cameras,images,points3D}.bin + <root>/images/*.png, rendered from a hidden scene."""
    g = torch.Generator().manual_seed(seed)
    means = torch.randn(n_gt, 3, generator=g) * torch.tensor([0.8, 0.5, 0.8])
    lo, hi = math.log(0.01), math.log(0.08)
    sc = dict(means=means, quats=torch.nn.functional.normalize(torch.randn(n_gt, 4, generator=g), dim=-1),
              scales=torch.exp(torch.rand(n_gt, 3, generator=g) * (hi - lo) + lo), opacities=torch.rand(n_gt, generator=g) * 0.7 + 0.25,
              sh=torch.cat([(torch.rand(n_gt, 1, 3, generator=g) - 0.5) * 2.5, (torch.rand(n_gt, 15, 3, generator=g) - 0.5) * 0.2], 1), sh_degree=3)
    hidden = scenes.to_splat_data(sc, dev)
    focal = 0.75 * width
    K = scenes.intrinsics(focal, focal, width / 2.0, height / 2.0)
    sp = os.path.join(root, "sparse", "0")
    os.makedirs(sp, exist_ok=True)
    os.makedirs(os.path.join(root, "images"), exist_ok=True)
    with open(os.path.join(sp, "cameras.bin"), "wb") as f:   # one PINHOLE camera (model id 1)
        f.write(struct.pack("<Q", 1) + struct.pack("<IiQQ", 1, 1, width, height) + struct.pack("<4d", focal, focal, width / 2.0, height / 2.0))
    bg = torch.zeros(3, device=dev)
    with open(os.path.join(sp, "images.bin"), "wb") as f:
        f.write(struct.pack("<Q", n_cams))
        for i in range(n_cams):
            a = 2 * math.pi * i / n_cams
            eye = (4.0 * math.sin(a), -0.8 - 0.6 * (i % 3), -4.0 * math.cos(a))
            vm = scenes.look_at_viewmat(eye, (0.0, 0.0, 0.0))
            name = "view_%03d.png" % i
            q, t = rotmat2qvec(vm[:3, :3].double().numpy()), vm[:3, 3].double().numpy()
            f.write(struct.pack("<I4d3dI", i + 1, *q, *t, 1) + name.encode() + b"\0" + struct.pack("<Q", 0))
            with torch.no_grad():
                img = rasterizer.rasterize_fused(rasterizer.Camera(viewmat=vm.to(dev), K=K.to(dev), width=width, height=height), hidden, bg).image
            io_image.save_image(os.path.join(root, "images", name), img)
    pick = torch.randperm(n_gt, generator=g)[:n_init]
    pts = (means[pick] + 0.01 * torch.randn(n_init, 3, generator=g)).double().numpy()
    rgb = ((sc["sh"][pick, 0] * 0.28209479177387814 + 0.5).clamp(0, 1) * 255).numpy().astype(np.uint8)
    with open(os.path.join(sp, "points3D.bin"), "wb") as f:
        f.write(struct.pack("<Q", n_init))
        for i in range(n_init):
            f.write(struct.pack("<Q3d3BdQ", i + 1, *pts[i], int(rgb[i, 0]), int(rgb[i, 1]), int(rgb[i, 2]), 0.5, 0))
    return root


def run(args):
    dev = "cuda:0"
    params = parameters.OptimizationParameters.from_file(args.config) if args.config else parameters.OptimizationParameters.preset("default")
    params.strategy = "mcmc"   # the only densification that gets a signal under --gut
    if args.iter:
        params.iterations = args.iter
    t_load = time.perf_counter()
    scene = io_colmap.load_colmap(args.data, args.images, device=dev)
    if scene.points is None or len(scene.points) == 0:
        raise SystemExit("the sparse model has no points3D: nothing to initialise the Gaussians from")
    images = scene.load_images(dev, res_div=args.resize_factor, max_width=args.max_width)
    cams = []
    for c, im in zip(scene.cameras, images):   # the decoded (possibly resized) size rules, as Camera::load_and_get_image sets _image_width / _image_height
        cam = c.camera
        h, w = int(im.shape[1]), int(im.shape[2])
        if (w, h) != (cam.width, cam.height):
            sx, sy = w / cam.width, h / cam.height
            Kc = cam.K.clone()
            Kc[0, 0] *= sx; Kc[0, 2] *= sx; Kc[1, 1] *= sy; Kc[1, 2] *= sy   # noqa: E702  (Camera::get_intrinsics, camera.cpp:94-103)
            cam = rasterizer.Camera(viewmat=cam.viewmat, K=Kc, width=w, height=h, camera_model=cam.camera_model, radial=cam.radial, tangential=cam.tangential)
        cams.append(cam)
    test_idx = [i for i in range(len(cams)) if args.eval and i % args.test_every == 0]   # src/training/dataset.hpp:42
    train_idx = [i for i in range(len(cams)) if i not in set(test_idx)]
    model, scene_scale = io_colmap.init_model_from_pointcloud(scene.points, scene.colors, scene.scene_center, sh_degree=params.sh_degree,
                                                             init_scaling=params.init_scaling, init_opacity=params.init_opacity, device=dev)
    bg = torch.zeros(3, device=dev)
    tr = trainer.Trainer(model, [cams[i] for i in train_idx], [images[i] for i in train_idx], params, bg, scene_scale=scene_scale, seed=0)
    t_load = time.perf_counter() - t_load
    ev_idx = test_idx if test_idx else train_idx[::8]
    ev = lambda: metrics.evaluate(model, [cams[i] for i in ev_idx], [images[i] for i in ev_idx], bg)  # noqa: E731
    before = ev()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train(params.iterations, log_every=args.log_every)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    after = ev()
    res = {"workload": "COLMAP capture %s (%s): %d training / %d evaluation images, %d x %d (first image), %d initial points" %
                       (os.path.abspath(args.data), args.images, len(train_idx), len(ev_idx), cams[0].width, cams[0].height, len(scene.points)),
           "iterations": params.iterations, "seconds": round(dt, 2), "iters_per_s": round(params.iterations / dt, 1), "load_seconds": round(t_load, 2),
           "gaussians_start": int(len(scene.points)), "gaussians_end": int(model.means.shape[0]), "max_cap": params.max_cap,
           "evaluated_on": "held-out test split (every %d-th image)" % args.test_every if test_idx else "every 8th training image",
           "psnr_before": round(before["psnr"], 2), "psnr_after": round(after["psnr"], 2), "ssim_before": round(before["ssim"], 4),
           "ssim_after": round(after["ssim"], 4), "active_sh_degree": model.active_sh_degree,
           "iterations_repeated_for_list_capacity": int(tr.capacity_misses),
           "non_finite_gaussians_relocated": int(getattr(tr.strategy, "nonfinite_relocated", 0)), "scene_scale": round(float(scene_scale), 4)}
    if args.output:
        os.makedirs(args.output, exist_ok=True)
        res["ply"] = io_ply.save_ply(model, os.path.join(args.output, "splat_%d.ply" % params.iterations))
    line = json.dumps(res)
    print(line)
    if args.json:
        with open(args.json, "w") as f:
            f.write(line + "\n")
    return res


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("-d", "--data", help="capture directory (sparse/0 or sparse + the images folder)")
    ap.add_argument("--images", default="images", help="images folder; a suffix _<k> scales the intrinsics by 1/k (images_4 ...)")
    ap.add_argument("--iter", type=int, default=0, help="iterations (default: the parameter file's, 30000)")
    ap.add_argument("--config", default=None, help="optimization parameter JSON (default: the reference's default_optimization_params.json values)")
    ap.add_argument("--eval", action="store_true", help="hold out every --test-every-th image and report PSNR / SSIM on it")
    ap.add_argument("--test-every", type=int, default=8)
    ap.add_argument("--resize-factor", type=int, default=-1)
    ap.add_argument("--max-width", type=int, default=3840)
    ap.add_argument("-o", "--output", default=None, help="directory for the trained model's PLY")
    ap.add_argument("--json", default=None)
    ap.add_argument("--log-every", type=int, default=0)
    ap.add_argument("--make-synthetic", default=None, metavar="DIR", help="write a small synthetic capture into DIR and exit")
    args = ap.parse_args()
    if args.make_synthetic:
        print(make_synthetic(args.make_synthetic))
        return
    if not args.data:
        ap.error("-d / --data is required")
    run(args)


if __name__ == "__main__":
    main()
