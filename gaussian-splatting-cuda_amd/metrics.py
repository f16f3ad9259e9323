"""Evaluation metrics of the reference (src/training/metrics/metrics.cpp:39-115): PSNR and SSIM.  LPIPS needs the pretrained
VGG weights file the reference loads at run time (weights/lpips_vgg.pt) and is not provided here.

PSNR keeps the reference's reduction: MSE per entry of the leading dimension, so an unbatched [3,H,W] image yields the mean of
three per-channel PSNRs (metrics.cpp:51-58).  SSIM is the mean of the "same"-padded 11x11 Gaussian-window SSIM map (metrics.cpp:
68-115) — exactly what the fused SSIM kernel computes with train = false."""
import torch

from . import loss


def psnr(pred, target, data_range=1.0):
    if pred.shape != target.shape:
        raise ValueError("Prediction and target must have the same shape")
    mse = (pred.contiguous() - target.contiguous()).pow(2).reshape(pred.shape[0], -1).mean(1, keepdim=True).clamp_min(1e-10)
    return float((20.0 * torch.log10(data_range / mse.sqrt())).mean())


def ssim(pred, target):
    with torch.no_grad():
        return float(loss.fused_ssim(pred, target, "same", train=False))


def evaluate(model, cameras, images, background=None):
    """MetricsEvaluator::evaluate (metrics.cpp:400-470) without LPIPS: mean PSNR / SSIM of the rendered test views and the time
    per rendered image."""
    import time

    from . import rasterizer
    ps, ss = [], []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        for cam, gt in zip(cameras, images):
            img = rasterizer.rasterize_fused(cam, model, background).image
            ps.append(psnr(img, gt))
            ss.append(ssim(img, gt))
    torch.cuda.synchronize()
    n = max(1, len(ps))
    return {"psnr": sum(ps) / n, "ssim": sum(ss) / n, "elapsed_time_per_image_s": (time.perf_counter() - t0) / n, "num_images": len(ps)}
