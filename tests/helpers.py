"""Shared helpers for the parity tests: run the CPU oracle over a synthetic scene (numpy in / numpy out)."""
import numpy as np

from oracle import oracle

TILE = 16


def np32(t):
    return np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.float32)


def oracle_pipeline(scene, dtype=np.float32, frag_rel=None, v_render_colors=None, v_render_alphas=None,
                    isect_override=None, colors_override=None, cam=None):
    """projection -> SH (+0.5, clamp_min 0) -> intersect -> offsets -> blend fwd (-> blend bwd) with the oracle.
    `isect_override=(tile_offsets, flatten_ids)` lets the blend be tested on identical binning."""
    f = lambda k: np.ascontiguousarray(scene[k].numpy(), dtype=dtype)  # noqa: E731
    means, quats, scales, opac = f("means"), f("quats"), f("scales"), f("opacities")
    viewmat, K = f("viewmat")[None], f("K")[None]
    W, H = scene["width"], scene["height"]
    sh, deg = f("sh"), scene["sh_degree"]
    bg = None if scene.get("background") is None else f("background")[None]
    cam = dict(cam or {})   # optional: camera_model, shutter, viewmats1, radial, tangential, thin_prism
    for k in ("viewmats1", "radial", "tangential", "thin_prism"):
        if cam.get(k) is not None:
            cam[k] = np.ascontiguousarray(np.asarray(cam[k]), dtype=dtype)
    radii, means2d, depths, conics, _ = oracle.projection_ut(means, quats, scales, opac, viewmat, K, W, H, **cam)
    campos = np.linalg.inv(viewmat.astype(np.float64))[:, :3, 3].astype(dtype)
    dirs = means[None] - campos[:, None]
    masks = (radii > 0).all(-1)
    if colors_override is None:
        colors = oracle.sh_fwd(deg, dirs.reshape(-1, 3), sh, masks.reshape(-1)).reshape(1, -1, 3)
        colors = np.maximum(colors + dtype(0.5), dtype(0))
    else:
        colors = colors_override.astype(dtype)
    tw, th = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    if isect_override is None:
        tpg, ids, fl = oracle.intersect_tile(means2d.astype(np.float32), radii, depths.astype(np.float32), 1, TILE, tw, th, True)
        offsets = oracle.intersect_offset(ids, 1, tw, th)
    else:
        offsets, fl = isect_override
        tpg, ids = None, None
    res = oracle.rasterize_fwd(means, quats, scales, colors, opac[None], bg, None, W, H, TILE, viewmat, K, offsets, fl,
                               frag_rel=frag_rel, **cam)
    out = dict(radii=radii, means2d=means2d, depths=depths, conics=conics, dirs=dirs, masks=masks, colors=colors,
               tiles_per_gauss=tpg, isect_ids=ids, flatten_ids=fl, tile_offsets=offsets, renders=res[0], alphas=res[1],
               last_ids=res[2], fragile=res[3] if frag_rel is not None else None)
    if v_render_colors is not None:
        g = oracle.rasterize_bwd(means, quats, scales, colors, opac[None], bg, None, W, H, TILE, viewmat, K, offsets, fl,
                                 res[1], res[2], v_render_colors.astype(dtype), v_render_alphas.astype(dtype), **cam)
        out.update(v_means=g[0], v_quats=g[1], v_scales=g[2], v_colors=g[3], v_opacities=g[4])
    return out


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def parity_record(name, **values):
    """Parity numbers of the -m gpu tests are kept, not just asserted: one JSON line per record in gpurun_out/parity.jsonl (merged back
    from the GPU box by gpurun; summarised into the tracked profiles/parity_rNN.md by tools/parity_report.py) and on stdout (-s)."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rec = {"test": name}
    for k, v in values.items():
        rec[k] = (float(v) if isinstance(v, (float, np.floating)) else int(v) if isinstance(v, (int, np.integer)) else v)
    line = json.dumps(rec)
    print("PARITY " + line)
    try:
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "parity.jsonl"), "a") as f:
            f.write(line + "\n")
    except OSError:
        pass
    return rec
