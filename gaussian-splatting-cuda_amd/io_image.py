"""Image decode / encode for the dataset side of the path: load_image / Camera::load_and_get_image /
save_image of the reference (src/core/image_io.cpp:112-345, src/core/camera.cpp:101-140) on Pillow instead of OpenImageIO.

Same size rules: `res_div` in {1, 2, 4, 8} divides both sides (integer division, at least 1 px); `max_width` > 0 then bounds
the LONGER side, keeping the aspect ratio with the reference's integer arithmetic (image_io.cpp:152-161, 190-203); alpha is
dropped, 1- and 2-channel files are expanded to RGB.  The down-sampling filter is Pillow's box/bilinear reduction, not OIIO's
resampler: resized pixels agree only approximately with the reference's (full-resolution loads are bit-identical)."""
import numpy as np
import torch
from PIL import Image


def target_size(w, h, res_div=1, max_width=0):
    if res_div not in (-1, 0, 1, 2, 4, 8):
        raise ValueError(f"load_image: unsupported resize factor {res_div}")
    div = res_div if res_div > 1 else 1
    nw, nh = max(1, w // div), max(1, h // div)
    if max_width > 0 and (nw > max_width or nh > max_width):
        if nw > nh:
            nw, nh = max(1, max_width), max(1, max_width * nh // nw)
        else:
            nw, nh = max(1, max_width * nw // nh), max(1, max_width)
    return nw, nh


def load_image(path, res_div=1, max_width=0):
    """-> uint8 array [H, W, 3]."""
    try:
        im = Image.open(path)
        im.load()
    except Exception as e:  # noqa: BLE001
        raise RuntimeError(f"Load failed: {path} : {e}") from e
    im = im.convert("RGB")  # drops alpha, expands grey (+alpha) to RGB
    w, h = im.size
    nw, nh = target_size(w, h, res_div, max_width)
    if (nw, nh) != (w, h):
        im = im.resize((nw, nh), Image.Resampling.BOX if (w % nw == 0 and h % nh == 0) else Image.Resampling.BILINEAR, reducing_gap=None)
    return np.ascontiguousarray(np.asarray(im, dtype=np.uint8))


def load_and_get_image(path, res_div=1, max_width=0, device="cpu"):
    """Camera::load_and_get_image: float32 [3, H, W] in [0, 1] on `device`."""
    a = torch.from_numpy(load_image(path, res_div, max_width))
    if str(device) != "cpu":
        a = a.pin_memory().to(device, non_blocking=True)
    return a.permute(2, 0, 1).to(torch.float32) / 255.0


def save_image(path, image):
    """save_image (image_io.cpp:263-345): [C,H,W] / [H,W,C] / [B,C,H,W] (first image) float in [0,1] -> 8-bit file."""
    t = image.detach().float().cpu()
    if t.dim() == 4:
        t = t[0]
    if t.dim() == 3 and t.shape[0] in (1, 3, 4) and t.shape[2] not in (1, 3, 4):
        t = t.permute(1, 2, 0)
    a = (t.clamp(0, 1) * 255.0).round().to(torch.uint8).numpy()
    if a.ndim == 3 and a.shape[2] == 1:
        a = a[:, :, 0]
    Image.fromarray(a).save(path)
    return path
