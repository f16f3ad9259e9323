"""Photometric loss of the training step.

`fused_ssim` mirrors the reference operator (include/kernels/fused_ssim.cuh:26-122: [N,C,H,W] or [C,H,W] images,
padding "same" | "valid", scalar mean of the map).  `photometric_loss` is Trainer::compute_photometric_loss
(src/training/trainer.cpp:103-127) evaluated straight on the blend's [C,H,W,3] output by the fused loss kernels
(csrc/gsx_ssim.hip): clamp, permute, L1, SSIM, crop and mean in one forward kernel; the backward writes
v_render_colors for the blend's backward."""
import torch

from . import ops

_C1, _C2 = 0.01 * 0.01, 0.03 * 0.03


class _FusedSSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img1, img2, padding, train):
        img1, img2 = img1.contiguous(), img2.contiguous()
        if img1.dim() == 3:
            img1 = img1.unsqueeze(0)
        if img2.dim() == 3:
            img2 = img2.unsqueeze(0)
        if img1.dim() != 4 or img2.dim() != 4:
            raise ValueError(f"fused_ssim expects 4D tensors [N,C,H,W], got img1: {img1.dim()}D, img2: {img2.dim()}D")
        if img1.shape != img2.shape:
            raise ValueError("img1 and img2 must have the same shape")
        smap, dm1, ds1, ds12 = ops.fusedssim(_C1, _C2, img1, img2, train)
        h, w = smap.shape[2], smap.shape[3]
        ctx.crop = padding == "valid" and h > 10 and w > 10
        ctx.valid = padding == "valid"
        if ctx.crop:
            smap = smap[:, :, 5:h - 5, 5:w - 5]
        ctx.save_for_backward(img1.detach(), img2, dm1, ds1, ds12)
        return smap

    @staticmethod
    def backward(ctx, grad_out):
        img1, img2, dm1, ds1, ds12 = ctx.saved_tensors
        dL_dmap = grad_out
        if ctx.valid:
            full = torch.zeros_like(img1)
            if ctx.crop:
                h, w = full.shape[2], full.shape[3]
                full[:, :, 5:h - 5, 5:w - 5] = dL_dmap
            dL_dmap = full
        return ops.fusedssim_backward(_C1, _C2, img1, img2, dL_dmap.contiguous(), dm1, ds1, ds12), None, None, None


def fused_ssim(img1, img2, padding="same", train=True):
    if padding not in ("same", "valid"):
        raise ValueError(f'fused_ssim: padding must be "same" or "valid" (got "{padding}")')
    return _FusedSSIM.apply(img1.contiguous(), img2, padding, train).mean()


def photometric_loss_reference(image_chw, gt_chw, lambda_dssim=0.2):
    """The reference's composition, op by op (trainer.cpp:103-127), on the clamped [3,H,W] image."""
    rendered = image_chw.unsqueeze(0) if image_chw.dim() == 3 else image_chw
    gt = gt_chw.unsqueeze(0) if gt_chw.dim() == 3 else gt_chw
    if rendered.shape != gt.shape:
        raise ValueError(f"ERROR: size mismatch – rendered {tuple(rendered.shape)} vs. ground truth {tuple(gt.shape)}")
    l1 = torch.nn.functional.l1_loss(rendered, gt)
    ssim_loss = 1.0 - fused_ssim(rendered, gt, "valid", True)
    return (1.0 - lambda_dssim) * l1 + lambda_dssim * ssim_loss


class _PhotometricLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, render_hwc, gt_chw, lambda_dssim):
        render_hwc = render_hwc.contiguous()
        loss3, ws = ops.photometric_loss_fwd(render_hwc, gt_chw, lambda_dssim)
        ctx.save_for_backward(render_hwc, gt_chw, ws)
        ctx.lambda_dssim = lambda_dssim
        ctx.mark_non_differentiable(loss3)
        ctx.set_materialize_grads(False)
        return loss3[0], loss3

    @staticmethod
    def backward(ctx, grad_loss, _grad_parts):
        if grad_loss is None:
            return None, None, None
        render_hwc, gt_chw, ws = ctx.saved_tensors
        return ops.photometric_loss_bwd(render_hwc, gt_chw, ws, ctx.lambda_dssim, grad_loss, 1.0), None, None


_ONES = {}


def backward(loss):
    """loss.backward() with a cached unit gradient: autograd otherwise launches a fill kernel for the root's ones_like(loss) on the step's
    critical path (5 us of a 1.35 ms iteration)."""
    one = _ONES.get(loss.device)
    if one is None or one.dtype != loss.dtype:
        one = _ONES[loss.device] = torch.ones((), dtype=loss.dtype, device=loss.device)
    loss.backward(gradient=one)


def photometric_loss(render_hwc, gt_chw, lambda_dssim=0.2, return_parts=False):
    """render_hwc: the blend's [C,H,W,3] (or [H,W,3]) unclamped output (RenderOutput.render_hwc); gt_chw [C,3,H,W] or [3,H,W]."""
    if render_hwc.dim() == 3:
        render_hwc = render_hwc.unsqueeze(0)
    if gt_chw.dim() == 3:
        gt_chw = gt_chw.unsqueeze(0)
    if render_hwc.shape[0] != gt_chw.shape[0] or render_hwc.shape[1:3] != gt_chw.shape[2:] or render_hwc.shape[3] != 3 or gt_chw.shape[1] != 3:
        raise ValueError(f"ERROR: size mismatch – rendered {tuple(render_hwc.shape)} vs. ground truth {tuple(gt_chw.shape)}")
    loss, parts = _PhotometricLoss.apply(render_hwc, gt_chw.contiguous(), float(lambda_dssim))
    return (loss, parts) if return_parts else loss
