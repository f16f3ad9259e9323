"""VERDICT r05 missing #4 / next #6: how many of a frame's tile intersections do the blend kernels actually READ?  SURVEY §0 finding 7 asks for the
algorithmic bytes on *processed* intersections next to all I.  A -DGSX_STATS build of the blend translation unit (bash tools/build_variant.sh stats
-DGSX_STATS, on the CPU side) counts them in the kernels: the forward pair kernel adds every chunk it stages per wave (two waves per tile: / 2 = list
entries staged before the tile's all-pixels-finished exit), the Gaussian-major backward every super-chunk it stages (entries at or before the tile's
last contributing id).  One fused render + backward of the scene's own camera; writes the `processed` entry of profiles/pmc.json (bench.py reports it
while the blend sources still hash to the value stored there).      python tools/processed_isects.py [1m|5m] [--write]"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GSX_TEST_SWITCHES", "1")
import torch  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "1m"
    lib = ctypes.CDLL(os.environ.get("GSX_STATS_LIB", os.path.join(ROOT, "tools", "variants", "libgsx_stats.so")), mode=ctypes.RTLD_GLOBAL)   # preloaded under the same soname: the extension binds to it
    import gsx  # noqa: F401
    from gsx import layout, rasterizer, scenes
    dev = "cuda:0"
    scene = {"1m": scenes.scene_1m, "5m": scenes.scene_5m}[which]()
    order = layout.morton_order(scene["means"])                     # bench.py's memory order
    for k in ("means", "quats", "scales", "opacities", "sh"):
        scene[k] = scene[k][order].contiguous()
    model = scenes.to_splat_data(scene, dev)
    for p in model.params():
        p.requires_grad_(True)
    cam = rasterizer.Camera(viewmat=scene["viewmat"].to(dev), K=scene["K"].to(dev), width=scene["width"], height=scene["height"])
    bg = scene["background"].to(dev)
    g = torch.Generator(device=dev).manual_seed(3)
    v = torch.randn(1, scene["height"], scene["width"], 3, device=dev, generator=g)
    buf = (ctypes.c_ulonglong * 16)()
    for it in range(2):            # the first render warms the capacity hint / kernel choice; the second one is counted
        lib.gsx_debug_read_stats(buf, 1)
        out = rasterizer.rasterize_fused(cam, model, bg)
        (out.render_hwc * v).sum().backward()
        torch.cuda.synchronize()
    lib.gsx_debug_read_stats(buf, 1)
    n_is = int(out.n_isects)
    sat = float((out.alpha.detach() > 0.9998).float().mean())
    rec = {"n_isects": n_is, "fwd_staged_entries": int(buf[1]) // 2, "bwd_staged_entries": int(buf[11]), "fwd_steps": int(buf[0]),
           "bwd_list_entries_4x4_blocks": int(buf[8]), "bwd_passes": int(buf[9]), "pixels_saturated_frac": round(sat, 4),
           "what": "entries of the tile lists the kernels staged (-DGSX_STATS counters of one fused render + backward, the scene's own camera, Morton order): forward = "
                   "chunks staged per wave / 2 waves (its early exit: all pixels of the tile finished, tested once per chunk of 64); backward = super-chunks staged "
                   "(entries at or before the tile's last contributing id)"}
    rec["fwd_processed_frac"] = round(rec["fwd_staged_entries"] / max(1, n_is), 4)
    rec["bwd_processed_frac"] = round(rec["bwd_staged_entries"] / max(1, n_is), 4)
    print(json.dumps(rec))
    if "--write" in sys.argv:
        path = os.path.join(ROOT, "profiles", "pmc.json")
        data = json.load(open(path)) if os.path.exists(path) else {}
        key = {"1m": "s1m_1080p", "5m": "s5m_4k"}[which]
        data.setdefault(key, {})["processed"] = rec
        json.dump(data, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
