"""where does the drop-in's day-one frame go?  Runs the reference's OWN gs::training::rasterize (+ its autograd
functions; rasterizer.cpp / rasterizer_autograd.cpp compiled unmodified: oracle/build_ref_callers.sh) on the gsx drop-in, forward + backward at S-1M,
n frames without a synchronisation in between (TEST INFRASTRUCTURE: uses oracle/_ref).

    python tools/dropin_trace.py [n] [gsx|ref|fused]            -> ms per frame (pipelined) on stdout
    rocprofv3 --kernel-trace --stats -d out -o dt --output-format csv -- python tools/dropin_trace.py 20 gsx ; python tools/dropin_trace.py --summarise out 20
"""
import csv
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def summarise(out, n_frames):
    f = glob.glob(out + "/**/*kernel_stats.csv", recursive=True)
    rows = list(csv.DictReader(open(f[0])))
    groups = {}
    for r in rows:
        nm = r["Name"]
        g = "gsx kernels (libgsx.so)" if ("gsx" in nm or "raster_" in nm or "bin_" in nm or "tile_sort" in nm or "sh_" in nm or "projection_ut" in nm or "pack_records" in nm
                                          or "isect_" in nm or "frontend" in nm or "splat_act" in nm) else "torch kernels launched by the reference's glue"
        e = groups.setdefault(g, dict(ns=0.0, calls=0, names={}))
        e["ns"] += float(r["TotalDurationNs"])
        e["calls"] += int(r["Calls"])
        e["names"][nm] = (float(r["TotalDurationNs"]), int(r["Calls"]))
    for g, e in groups.items():
        print("## %s: %.3f ms per frame, %.1f launches per frame" % (g, e["ns"] / 1e6 / n_frames, e["calls"] / n_frames))
        for nm, (ns, c) in sorted(e["names"].items(), key=lambda kv: -kv[1][0])[:14]:
            print("  %8.1f us/frame  %5.1f launches/frame  %s" % (ns / 1e3 / n_frames, c / n_frames, nm[:120]))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--summarise":
        return summarise(sys.argv[2], int(sys.argv[3]))
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    flavour = sys.argv[2] if len(sys.argv) > 2 else "gsx"
    import torch
    import gsx  # noqa: F401
    from gsx import rasterizer, scenes
    from oracle import ref_callers
    DEV = "cuda:0"
    sc = scenes.scene_1m()
    if os.environ.get("GSX_TRACE_MORTON", "0") == "1":   # the bench's memory order (gsx.layout); default: the generator's order = a model as the reference leaves it
        from gsx import layout
        order = layout.morton_order(sc["means"])
        for k in ("means", "quats", "scales", "opacities", "sh"):
            sc[k] = sc[k][order].contiguous()
    H, W = sc["height"], sc["width"]
    g = torch.Generator(device=DEV).manual_seed(5)
    v_img, v_alpha = torch.randn(3, H, W, device=DEV, generator=g), torch.randn(1, H, W, device=DEV, generator=g)
    if flavour == "fused":
        model = scenes.to_splat_data(sc, DEV)
        for p in model.params():
            p.requires_grad_(True)
        cam = rasterizer.Camera(viewmat=sc["viewmat"].to(DEV), K=sc["K"].to(DEV), width=W, height=H)
        bg = sc["background"].to(DEV)
        v_hwc = v_img.permute(1, 2, 0)[None].contiguous()

        def frame():
            for p in model.params():
                p.grad = None
            out = rasterizer.rasterize_fused(cam, model, bg)
            ((out.render_hwc * v_hwc).sum() + (out.alpha * v_alpha).sum()).backward()
    else:
        mod = ref_callers.load(flavour)
        op = sc["opacities"].clamp(1e-6, 1 - 1e-6)
        P = dict(means=sc["means"], sh0=sc["sh"][:, :1].contiguous(), shN=sc["sh"][:, 1:].contiguous(), scaling_raw=torch.log(sc["scales"]),
                 rotation_raw=sc["quats"], opacity_raw=torch.logit(op).unsqueeze(-1))
        P = {k: v.to(DEV).clone().requires_grad_(True) for k, v in P.items()}
        vm, K = sc["viewmat"], sc["K"]
        R, T, bg = vm[:3, :3].contiguous(), vm[:3, 3].contiguous(), sc["background"].to(DEV)
        e0 = torch.empty(0)

        def frame():
            for t in P.values():
                t.grad = None
            img, alpha, radii = mod.render(P["means"], P["sh0"], P["shN"], P["scaling_raw"], P["rotation_raw"], P["opacity_raw"], sc["sh_degree"], R, T,
                                           float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), W, H, bg, e0, e0, 0)
            ((img * v_img).sum() + (alpha * v_alpha).sum()).backward()
    for _ in range(3):
        frame()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        frame()
    torch.cuda.synchronize()
    print("%s: %.4f ms per frame (forward + backward, %d frames, no synchronisation between frames)" % (flavour, (time.perf_counter() - t0) / n * 1e3, n))


if __name__ == "__main__":
    main()
