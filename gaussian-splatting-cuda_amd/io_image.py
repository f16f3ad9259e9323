"""Image decode / encode for the dataset side of the path: load_image / Camera::load_and_get_image /
save_image of the reference (src/core/image_io.cpp:112-345, src/core/camera.cpp:101-140) on Pillow instead of OpenImageIO.

Same size rules: `res_div` in {1, 2, 4, 8} divides both sides (integer division, at least 1 px); `max_width` > 0 then bounds
the LONGER side, keeping the aspect ratio with the reference's integer arithmetic (image_io.cpp:152-161, 190-203); alpha is
dropped, 1- and 2-channel files are expanded to RGB.  Down-sampling restates the reference's call
`OIIO::ImageBufAlgo::resample(dst, src, /*interpolate=*/true)` (image_io.cpp:33-49): OpenImageIO is a vcpkg dependency that is not in
this image, so `resample_oiio` follows its published algorithm (imagebufalgo_xform.cpp `resample_`, imagebuf.cpp `interppixel_`,
fmath.h `bilerp`) in float32 and is pinned by hand-computed fixtures (tests/test_io.py), not against the library."""
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


def resample_oiio(src, nw, nh):
    """OIIO::ImageBufAlgo::resample(dst, src, interpolate=true) for uint8 [H,W,C] -> [nh,nw,C], in the library's float32 arithmetic:
    destination pixel (x, y) samples the source at the position of its CENTRE, ((x + 0.5) / nw * w, (y + 0.5) / nh * h), by bilinear
    interpolation between the four texel centres around it (interppixel: position - 0.5, floor / frac, texels outside the image are
    black); texels are read as v * (1/255) and the result is stored as (uint8)(clamp(f * 255 + 0.5))."""
    src = np.ascontiguousarray(src, dtype=np.uint8)
    h, w = src.shape[:2]
    f32 = np.float32
    srcf = src.astype(f32) * f32(1.0 / 255.0)

    def axis(n_dst, n_src):
        s = (np.arange(n_dst, dtype=f32) + f32(0.5)) * (f32(1.0) / f32(n_dst))      # (x - dstfx + 0.5) * dstpixelwidth
        pos = s * f32(n_src) - f32(0.5)                                             # srcfx + s * srcfw, then interppixel's -0.5
        tex = np.floor(pos)
        return tex.astype(np.int64), (pos - tex).astype(f32)

    xt, xf = axis(nw, w)
    yt, yf = axis(nh, h)

    def texel(yy, xx):   # WrapBlack: zero outside the data window
        ok = ((yy >= 0) & (yy < h))[:, None] & ((xx >= 0) & (xx < w))[None, :]
        v = srcf[np.clip(yy, 0, h - 1)[:, None], np.clip(xx, 0, w - 1)[None, :]]
        return np.where(ok[..., None], v, f32(0.0))

    v0, v1, v2, v3 = texel(yt, xt), texel(yt, xt + 1), texel(yt + 1, xt), texel(yt + 1, xt + 1)
    s, t = xf[None, :, None], yf[:, None, None]
    s1, t1 = f32(1.0) - s, f32(1.0) - t
    out = t1 * (s1 * v0 + s * v1) + t * (s1 * v2 + s * v3)                           # fmath.h bilerp
    return np.clip(out * f32(255.0) + f32(0.5), f32(0.0), f32(255.0)).astype(np.uint8)


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
    a = np.array(im, dtype=np.uint8)   # (a writable copy: torch.from_numpy refuses to promise anything about PIL's read-only buffer)
    if (nw, nh) != (w, h):
        a = resample_oiio(a, nw, nh)
    return a


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
