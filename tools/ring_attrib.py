"""the S-8cam ring cameras where the blend backward of HIP and of the reference kernel are more than 1e-3 apart
(cameras 1, 3, 5; 7 as a control).  TEST INFRASTRUCTURE (uses oracle/ and oracle/_ref): runs on a GPU box.

Per camera, on the reference chain's colours and lists (identical inputs for everybody):
  * forward of: the reference kernel, HIP fast (Delta-form), HIP reference-order (GSX_RASTER_PATH=generic), the float64 oracle;
  * backward of each of the four in TWO settings:
      "mixed" = on the REFERENCE kernel's forward state (render_alphas, last_ids) — what the stage-wise test feeds everybody;
      "own"   = on the implementation's OWN forward state — the gradient of the function that implementation actually rendered;
  * rel-L2 of every fp32 backward against the float64 backward of the same setting, and against the reference kernel;
  * alpha: pixels whose |d alpha| > 1e-4 against the reference kernel, explained by the oracle's threshold flags / last-id differences or priced against float64.
Attribution (camera 3 unless --attrib-cam): per-Gaussian squared error of v_scales / v_quats against float64 (own setting): share of the top-100
Gaussians, and what they look like (scale ratio, depth, opacity, radius, tiles, position in their tile's list).

    python tools/ring_attrib.py [--cams 1,3,5,7] [--attrib-cam 3] > gpurun_out/ring_attrib.jsonl
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GSX_TEST_SWITCHES", "1")
import numpy as np  # noqa: E402
import torch  # noqa: E402

import gsx  # noqa: E402,F401
from gsx import ops, scenes  # noqa: E402
from oracle import oracle, ref_hip  # noqa: E402

DEV = "cuda:0"
GRADS = ["v_means", "v_quats", "v_scales", "v_colors", "v_opacities"]


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64).reshape(-1), np.asarray(b, np.float64).reshape(-1)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def dev(a):
    return a.to(DEV).contiguous() if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def np32(t):
    return t.detach().cpu().numpy().astype(np.float32)


def emit(what, **kw):
    print(json.dumps(dict(what=what, **kw)), flush=True)


def fp32_camera_frame(vm):
    """numpy float32 restatement of gsx_record.hpp: make_cam_frame = the reference's pose round trip (Cameras.cuh:42-52,258-262): (R_inv [3,3], camera centre [3])."""
    f = np.float32
    se3 = vm.astype(f).reshape(-1)
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = se3[0], se3[1], se3[2], se3[4], se3[5], se3[6], se3[8], se3[9], se3[10]
    cands = [f(m00 + m11 + m22), f(m00 - m11 - m22), f(m11 - m00 - m22), f(m22 - m00 - m11)]
    bi, big = 0, cands[0]
    for k in (1, 2, 3):
        if cands[k] > big:
            bi, big = k, cands[k]
    bv = f(np.sqrt(f(big + f(1))) * f(0.5))
    mult = f(f(0.25) / bv)
    q = {0: (bv, (m21 - m12) * mult, (m02 - m20) * mult, (m10 - m01) * mult), 1: ((m21 - m12) * mult, bv, (m10 + m01) * mult, (m02 + m20) * mult),
         2: ((m02 - m20) * mult, (m10 + m01) * mult, bv, (m21 + m12) * mult), 3: ((m10 - m01) * mult, (m02 + m20) * mult, (m21 + m12) * mult, bv)}[bi]
    w, x, y, z = [f(v) for v in q]
    d = f(x * x + y * y + z * z + w * w)
    w, x, y, z = f(w / d), f(-x / d), f(-y / d), f(-z / d)
    xx, yy, zz, xz, xy, yz, wx, wy, wz = f(x * x), f(y * y), f(z * z), f(x * z), f(x * y), f(y * z), f(w * x), f(w * y), f(w * z)
    R = np.array([[f(1) - f(2) * f(yy + zz), f(2) * f(xy - wz), f(2) * f(xz + wy)], [f(2) * f(xy + wz), f(1) - f(2) * f(xx + zz), f(2) * f(yz - wx)],
                  [f(2) * f(xz - wy), f(2) * f(yz + wx), f(1) - f(2) * f(xx + yy)]], dtype=f)
    t = np.array([se3[3], se3[7], se3[11]], dtype=f)
    rt = np.array([f(f(R[i, 0] * t[0]) + f(R[i, 1] * t[1])) + f(R[i, 2] * t[2]) for i in range(3)], dtype=f)
    return R, -rt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cams", default="1,3,5,7")
    ap.add_argument("--attrib-cam", type=int, default=3)
    ap.add_argument("--top", type=int, default=100)
    args = ap.parse_args()
    ref = ref_hip.load()
    assert ref is not None, "oracle/_ref/gsplat_ref_hip.so not built"
    sc0 = scenes.scene_1m()
    W, H = sc0["width"], sc0["height"]
    rng = np.random.default_rng(3)
    v_rc_n = rng.standard_normal((1, H, W, 3)).astype(np.float32)
    v_ra_n = rng.standard_normal((1, H, W, 1)).astype(np.float32)
    v_rc, v_ra = dev(v_rc_n), dev(v_ra_n)
    ut = ops.UnscentedTransformParameters()
    f32 = lambda k: np.ascontiguousarray(sc0[k].numpy(), np.float32)  # noqa: E731
    f64 = lambda k: np.ascontiguousarray(sc0[k].numpy(), np.float64)  # noqa: E731
    for ci in [int(c) for c in args.cams.split(",")]:
        vm = scenes.ring_cameras(8)[ci]
        a = dict(means=dev(sc0["means"]), quats=dev(sc0["quats"]), scales=dev(sc0["scales"]), opacities=dev(sc0["opacities"]), sh=dev(sc0["sh"]),
                 viewmat=dev(vm[None]), K=dev(sc0["K"][None]), background=dev(sc0["background"][None]))
        R = ref_hip.render_chain(ref, a["means"], a["quats"], a["scales"], a["opacities"], a["sh"], 3, a["viewmat"], a["K"], W, H, a["background"],
                                 v_render_colors=v_rc, v_render_alphas=v_ra)
        torch.cuda.synchronize()
        op = a["opacities"][None].contiguous()
        common = (a["means"], a["quats"], a["scales"], R["colors"], op, a["background"], None, W, H, 16, a["viewmat"], None, a["K"], ops.CameraModelType.PINHOLE, ut,
                  ops.ShutterType.GLOBAL, None, None, None, R["tile_offsets"], R["flatten_ids"])
        off_n, fl_n = R["tile_offsets"].cpu().numpy(), R["flatten_ids"].cpu().numpy()
        col_n = np32(R["colors"])
        vm_n, K_n = vm[None].numpy(), sc0["K"][None].numpy()
        impl = {}
        impl["reference"] = dict(fwd=(np32(R["renders"]), np32(R["alphas"]), R["last_ids"].cpu().numpy()), own=[np32(R[g]) for g in GRADS], mixed=[np32(R[g]) for g in GRADS])
        for path in ("fast", "generic"):
            if path == "generic":
                os.environ["GSX_RASTER_PATH"] = "generic"
            try:
                G = ops.rasterize_to_pixels_from_world_3dgs_fwd(*common)
                Bm = ops.rasterize_to_pixels_from_world_3dgs_bwd(*common, R["alphas"], R["last_ids"], v_rc, v_ra)
                Bo = ops.rasterize_to_pixels_from_world_3dgs_bwd(*common, G[1], G[2], v_rc, v_ra)
                torch.cuda.synchronize()
            finally:
                os.environ.pop("GSX_RASTER_PATH", None)
            impl["hip_" + path] = dict(fwd=(np32(G[0]), np32(G[1]), G[2].cpu().numpy()), own=[np32(g) for g in Bo], mixed=[np32(g) for g in Bm])
        o64 = (f64("means"), f64("quats"), f64("scales"), col_n.astype(np.float64), f64("opacities")[None], f64("background")[None], None, W, H, 16,
               vm_n.astype(np.float64), K_n.astype(np.float64), off_n, fl_n)
        r64, a64, l64, frag = oracle.rasterize_fwd(*o64, frag_rel=4e-3)
        g64_own = oracle.rasterize_bwd(*o64, a64, l64, v_rc_n.astype(np.float64), v_ra_n.astype(np.float64))
        g64_mixed = oracle.rasterize_bwd(*o64, impl["reference"]["fwd"][1].astype(np.float64), impl["reference"]["fwd"][2], v_rc_n.astype(np.float64),
                                         v_ra_n.astype(np.float64))
        # ---- forward: RGB and alpha against the reference kernel and against float64
        r_ren, r_alp, r_last = impl["reference"]["fwd"]
        for name in ("hip_fast", "hip_generic", "reference"):
            ren, alp, last = impl[name]["fwd"]
            e_rgb64 = np.abs(ren.astype(np.float64) - r64).max(-1)[0]
            e_a64 = np.abs(alp.astype(np.float64) - a64)[0, ..., 0]
            rec = dict(camera=ci, impl=name, rgb_vs_f64_max=float(e_rgb64.max()), rgb_vs_f64_over_1e4=int((e_rgb64 > 1e-4).sum()), rgb_vs_f64_mean=float(e_rgb64.mean()),
                       alpha_vs_f64_max=float(e_a64.max()), alpha_vs_f64_over_1e4=int((e_a64 > 1e-4).sum()), alpha_vs_f64_mean=float(e_a64.mean()),
                       last_id_vs_f64_mismatch=int((last != l64).sum()))
            if name != "reference":
                e_rgb = np.abs(ren - r_ren).max(-1)[0]
                e_a = np.abs(alp - r_alp)[0, ..., 0]
                flagged = (frag[0] != 0) | (last[0] != r_last[0])
                over_a = e_a > 1e-4
                un = over_a & ~flagged
                rec.update(rgb_vs_ref_over_1e4=int((e_rgb > 1e-4).sum()), alpha_vs_ref_over_1e4=int(over_a.sum()), alpha_vs_ref_max=float(e_a.max()),
                           alpha_over_explained_by_flags=int((over_a & flagged).sum()), alpha_over_unexplained=int(un.sum()),
                           alpha_unexplained_max=float(e_a[un].max()) if un.any() else 0.0,
                           alpha_unexplained_hip_vs_f64_mean=float(e_a64[un].mean()) if un.any() else 0.0,
                           alpha_unexplained_ref_vs_f64_mean=float(np.abs(r_alp.astype(np.float64) - a64)[0, ..., 0][un].mean()) if un.any() else 0.0,
                           alpha_unexplained_hip_vs_f64_max=float(e_a64[un].max()) if un.any() else 0.0,
                           alpha_unexplained_ref_vs_f64_max=float(np.abs(r_alp.astype(np.float64) - a64)[0, ..., 0][un].max()) if un.any() else 0.0)
            emit("forward", **rec)
        # ---- backward: every fp32 implementation against float64 (same setting) and against the reference kernel
        for name in ("hip_fast", "hip_generic", "reference"):
            for setting, g64 in (("mixed", g64_mixed), ("own", g64_own)):
                if name == "reference" and setting == "mixed":
                    continue   # the reference kernel on its own forward state IS its "own" setting; priced against both float64 settings below
                got = impl[name][setting]
                rec = dict(camera=ci, impl=name, setting=setting)
                for n, g, o in zip(GRADS, got, g64):
                    rec[n + "_vs_f64"] = rel_l2(g, o)
                    rec[n + "_vs_reference_kernel"] = rel_l2(g, impl["reference"]["own"][GRADS.index(n)])
                emit("backward", **rec)
        emit("backward", camera=ci, impl="reference", setting="own forward state, priced against the float64 backward on the REFERENCE's forward state (mixed yardstick)",
             **{n + "_vs_f64": rel_l2(g, o) for n, g, o in zip(GRADS, impl["reference"]["own"], g64_mixed)})
        # Is the distance of EVERY fp32 evaluation to float64 on this camera the reference's fp32 pose round trip?  The float64 oracle once more, on a pose whose
        # translation is chosen so that ITS camera centre is the fp32 one (the 5e-6 shift of cameras 3 / 5; the 7e-7 non-orthonormality of the fp32 R_inv cannot be
        # handed to it through a view matrix): if the fp32 backwards are much closer to THIS float64 backward, the distance was the frame, not the kernels.
        _, c32 = fp32_camera_frame(vm.numpy())
        vm_c = vm.numpy().astype(np.float64).copy()
        vm_c[:3, 3] = -(vm_c[:3, :3] @ c32.astype(np.float64))
        o64c = o64[:10] + (vm_c[None],) + o64[11:]
        r64c, a64c, l64c, _ = oracle.rasterize_fwd(*o64c, frag_rel=4e-3)
        g64c = oracle.rasterize_bwd(*o64c, a64c, l64c, v_rc_n.astype(np.float64), v_ra_n.astype(np.float64))
        for name in ("hip_fast", "hip_generic", "reference"):
            ren, alp, last = impl[name]["fwd"]
            ea = np.abs(alp.astype(np.float64) - a64c)[0, ..., 0]
            emit("vs_float64_on_the_fp32_camera_centre", camera=ci, impl=name, alpha_over_1e4=int((ea > 1e-4).sum()), alpha_mean=float(ea.mean()),
                 **{n + "_vs_f64c": rel_l2(g, o) for n, g, o in zip(GRADS, impl[name]["own"], g64c)})
        emit("yardsticks", camera=ci, **{n + "_f64own_vs_f64mixed": rel_l2(a_, b_) for n, a_, b_ in zip(GRADS, g64_own, g64_mixed)})
        # ---- attribution: who carries the squared error of v_scales / v_quats?
        if ci == args.attrib_cam:
            sc_n, q_n, op_n, mu_n = f32("scales"), f32("quats"), f32("opacities"), f32("means")
            ratio = sc_n.max(1) / sc_n.min(1)
            depth = np32(R["depths"])[0]
            radii = R["radii"].cpu().numpy()[0].max(-1)
            tiles = R["tiles_per_gauss"].cpu().numpy().reshape(-1)
            # position of a Gaussian in the lists it appears in: mean of (index inside the tile's list) over its entries
            offs = np.concatenate([off_n.reshape(-1).astype(np.int64), [fl_n.size]])
            seg_len = np.diff(offs)
            seg_of = np.repeat(np.arange(seg_len.size), seg_len)
            pos = np.arange(fl_n.size) - offs[seg_of]
            pos_sum = np.bincount(fl_n, weights=pos, minlength=mu_n.shape[0])
            len_sum = np.bincount(fl_n, weights=seg_len[seg_of], minlength=mu_n.shape[0])
            cnt = np.maximum(np.bincount(fl_n, minlength=mu_n.shape[0]), 1)
            for gname in ("v_scales", "v_quats", "v_means"):
                k = GRADS.index(gname)
                for setting, g64 in (("own", g64_own), ("mixed", g64_mixed)):
                    tot64 = float((g64[k].astype(np.float64) ** 2).sum())
                    per = {}
                    for name in ("hip_fast", "hip_generic", "reference"):
                        got = impl[name]["own" if (name == "reference" or setting == "own") else "mixed"][k]
                        per[name] = ((got.astype(np.float64) - g64[k]) ** 2).reshape(got.shape[0], -1).sum(1)
                    for name, e in per.items():
                        order = np.argsort(-e)
                        top = order[:args.top]
                        emit("attribution", camera=ci, gradient=gname, setting=setting, impl=name, rel_l2=float(np.sqrt(e.sum() / tot64)),
                             top_share_of_squared_error=float(e[top].sum() / e.sum()), top1_share=float(e[order[0]] / e.sum()),
                             gaussians_for_half_of_the_error=int(np.searchsorted(np.cumsum(e[order]) / e.sum(), 0.5) + 1),
                             rel_l2_without_top=float(np.sqrt((e.sum() - e[top].sum()) / tot64)),
                             top_scale_ratio_median=float(np.median(ratio[top])), all_scale_ratio_median=float(np.median(ratio)),
                             top_depth_median=float(np.median(depth[top])), visible_depth_median=float(np.median(depth[radii > 0])),
                             top_radius_px_median=float(np.median(radii[top])), visible_radius_px_median=float(np.median(radii[radii > 0])),
                             top_opacity_median=float(np.median(op_n[top])), top_tiles_median=float(np.median(tiles[top])),
                             top_list_position_median=float(np.median(pos_sum[top] / cnt[top])), top_list_length_median=float(np.median(len_sum[top] / cnt[top])),
                             all_list_position_median=float(np.median((pos_sum / cnt)[tiles > 0])), all_list_length_median=float(np.median((len_sum / cnt)[tiles > 0])),
                             top_min_scale_median=float(np.median(sc_n.min(1)[top])), all_min_scale_median=float(np.median(sc_n.min(1))),
                             top_overlap_with_reference_top=int(np.intersect1d(top, np.argsort(-per["reference"])[:args.top]).size))
                    # is HIP's excess over the reference concentrated?  d = e_hip - e_ref per Gaussian
                    d = per["hip_fast"] - per["reference"]
                    od = np.argsort(-d)
                    emit("attribution_excess", camera=ci, gradient=gname, setting=setting, excess_total=float(d.sum()), excess_of_top=float(d[od[:args.top]].sum()),
                         deficit_of_bottom=float(d[od[-args.top:]].sum()), positive_part=float(d[d > 0].sum()), negative_part=float(d[d < 0].sum()),
                         reference_total=float(per["reference"].sum()), hip_total=float(per["hip_fast"].sum()))


if __name__ == "__main__":
    main()
