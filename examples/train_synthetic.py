"""End-to-end example on a GPU box: fit a perturbed Gaussian cloud to renders of the original with the `--gut` training step
(MCMC strategy, fused L1 + SSIM loss, fused Adam), report PSNR / SSIM, export and re-import the model as a 3DGS PLY.

    python examples/train_synthetic.py [iterations]
"""
import math
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import gsx  # noqa: E402,F401
from gsx import io_ply, metrics, rasterizer, scenes, strategy, trainer  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    dev = "cuda:0"
    sc = scenes.scene_small(seed=5, N=20000)
    gt = scenes.to_splat_data(sc, dev)
    W, H = sc["width"], sc["height"]
    cams = []
    for k in range(8):
        vm = sc["viewmat"].clone()
        vm[0, 3] += 0.2 * math.cos(k * math.pi / 4)
        vm[1, 3] += 0.2 * math.sin(k * math.pi / 4)
        cams.append(rasterizer.Camera(viewmat=vm.to(dev), K=sc["K"].to(dev), width=W, height=H))
    bg = sc["background"].to(dev)
    with torch.no_grad():
        images = [rasterizer.rasterize_fused(c, gt, bg).image.clone() for c in cams]
    g = torch.Generator().manual_seed(1)
    model = scenes.to_splat_data(dict(sc), dev)
    model.sh = (gt.sh + 0.3 * torch.randn(gt.sh.shape, generator=g).to(dev)).contiguous()
    model.means = (gt.means + 0.01 * torch.randn(gt.means.shape, generator=g).to(dev)).contiguous()
    params = strategy.OptimizationParameters(iterations=iters, start_refine=iters // 8, refine_every=max(1, iters // 8), stop_refine=iters,
                                             max_cap=22000)
    tr = trainer.Trainer(model, cams, images, params, bg, seed=0)

    def evaluate():
        r = metrics.evaluate(model, cams, images, bg)
        return r["psnr"], r["ssim"]

    print("before: PSNR %.2f dB  SSIM %.4f" % evaluate())
    tr.train(iters, log_every=max(1, iters // 4))
    print("after : PSNR %.2f dB  SSIM %.4f  (%d Gaussians)" % (evaluate() + (model.means.shape[0],)))
    with tempfile.TemporaryDirectory() as d:
        path = io_ply.save_ply(model, d, iteration=iters)
        back = io_ply.load_ply(path, dev)
        print("PLY round trip:", os.path.getsize(path), "bytes,", back.means.shape[0], "Gaussians, means identical:",
              bool(torch.equal(back.means, model.means.detach())))


if __name__ == "__main__":
    main()
