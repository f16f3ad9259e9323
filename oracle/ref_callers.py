"""TEST INFRASTRUCTURE ONLY — loader of oracle/_ref/gsplat_ref_callers_{gsx,ref}.so: the reference's OWN render call site
(gs::training::rasterize: src/training/rasterization/rasterizer.cpp + rasterizer_autograd.cpp, compiled unmodified by
oracle/build_ref_callers.sh) linked once against this repository's drop-in backend ("gsx") and once against the reference's own kernels
compiled for gfx950 ("ref").  Only tests/ may import this module.

    mod.render(means, sh0, shN, scaling_raw, rotation_raw, opacity_raw, sh_degree, R [3,3], T [3], fx, fy, cx, cy, width, height,
               bg [3], radial, tangential, camera_model) -> (image [3,H,W], alpha [1,H,W], radii [N])
The tensors are the RAW parameters (autograd leaves); the reference's SplatData getters activate them, so .backward() on the image
fills their .grad through the reference's autograd functions."""
import importlib.util
import os

HERE = os.path.dirname(os.path.abspath(__file__))
_cache = {}


def path(flavour):
    return os.path.join(HERE, "_ref", "gsplat_ref_callers_%s.so" % flavour)


def load(flavour):
    """flavour: "gsx" | "ref".  The extension module, or None when it has not been built (no /root/reference at build time)."""
    if flavour in _cache:
        return _cache[flavour]
    mod = None
    if os.path.exists(path(flavour)):
        import torch  # noqa: F401  (libtorch / libamdhip64 first)
        name = "gsplat_ref_callers_%s" % flavour
        spec = importlib.util.spec_from_file_location(name, path(flavour))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    _cache[flavour] = mod
    return mod
