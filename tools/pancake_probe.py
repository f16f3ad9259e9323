"""Ultra-thin large discs (one scale axis 1e-9 .. 1e-5, the others 0.2 .. 0.4: what 25 000 MCMC iterations with the scale regulariser leave behind, tools/soak_run.py):
non-finite values in the outputs of the blend forward / backward — HIP fast kernels, HIP reference-order kernels, the reference's own kernels.   python tools/pancake_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
if os.environ.get("GSX_VARIANT_LIB"):   # a variant build of libgsx.so preloaded under the same soname (tools/build_variant.sh)
    import ctypes
    ctypes.CDLL(os.environ["GSX_VARIANT_LIB"], mode=ctypes.RTLD_GLOBAL)
import gsx  # noqa: F401
from gsx import ops, scenes
from oracle import ref_hip
from tests.golden import ref_hip_cases

DEV = "cuda:0"
ref = ref_hip.load()
sc = ref_hip_cases.small_scene(scenes, N=3000)
g = torch.Generator().manual_seed(3)
N = 3000
lo_exp = float(sys.argv[1]) if len(sys.argv) > 1 else -9.5
pick = torch.randperm(N, generator=g)[:300]
sc["scales"][pick] = torch.rand(300, 3, generator=g) * 0.2 + 0.2
ax = torch.randint(0, 3, (300,), generator=g)
hi_exp = float(sys.argv[2]) if len(sys.argv) > 2 else -5.0
sc["scales"][pick, ax] = 10.0 ** (torch.rand(300, generator=g) * (hi_exp - lo_exp) + lo_exp)
sc["opacities"][pick] = torch.rand(300, generator=g) * 0.15 + 0.05
d = lambda x: x.to(DEV).contiguous()  # noqa: E731
means, quats, scales, opac = d(sc["means"]), d(sc["quats"]), d(sc["scales"]), d(sc["opacities"])
W, H = sc["width"], sc["height"]
vm, K, bg = d(sc["viewmat"][None]), d(sc["K"][None]), d(sc["background"][None])
R = ref_hip.render_chain(ref, means, quats, scales, opac, d(sc["sh"]), sc["sh_degree"], vm, K, W, H, bg)
rng = np.random.default_rng(5)
v_rc = torch.from_numpy(rng.standard_normal((1, H, W, 3)).astype(np.float32)).to(DEV)
v_ra = torch.from_numpy(rng.standard_normal((1, H, W, 1)).astype(np.float32)).to(DEV)
nf = lambda t: int((~torch.isfinite(t)).sum())  # noqa: E731
print("scene: %d visible, %d intersections, thin discs visible %d" % (int((R["radii"] > 0).all(-1).sum()), R["flatten_ids"].numel(), int((R["radii"][0][pick.to(DEV)] > 0).all(-1).sum())))
rargs = (means, quats, scales, R["colors"], opac[None].contiguous(), bg, None, W, H, 16, vm, None, K, 0, None, 4, None, None, None, R["tile_offsets"], R["flatten_ids"])
r_f = ref.rasterize_to_pixels_from_world_3dgs_fwd(*rargs)
r_g = ref.rasterize_to_pixels_from_world_3dgs_bwd(*rargs, r_f[1], r_f[2], v_rc, v_ra)
print("reference kernels : forward non-finite %d / %d, backward %s" % (nf(r_f[0]), nf(r_f[1]), [nf(x) for x in r_g]))
from oracle import oracle
f64 = lambda t: np.ascontiguousarray(t.detach().cpu().numpy(), np.float64)  # noqa: E731
o_f = oracle.rasterize_fwd(f64(means), f64(quats), f64(scales), f64(R["colors"]), f64(opac)[None], f64(bg), None, W, H, 16, f64(vm), f64(K), R["tile_offsets"].cpu().numpy(), R["flatten_ids"].cpu().numpy())
o_g = oracle.rasterize_bwd(f64(means), f64(quats), f64(scales), f64(R["colors"]), f64(opac)[None], f64(bg), None, W, H, 16, f64(vm), f64(K), R["tile_offsets"].cpu().numpy(), R["flatten_ids"].cpu().numpy(),
                           o_f[1], o_f[2], f64(v_rc), f64(v_ra))
pk = pick.numpy()
wide = np.ones((N, 3), bool); wide[pk, ax.numpy()] = False   # every scale axis except the discs' thin ones


def vs64(g, tag):
    rl = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))  # noqa: E731
    gs = [x.detach().cpu().numpy().astype(np.float64) for x in g]
    fin = np.isfinite(gs[2]).all(-1)
    print("      %s vs the float64 backward (rel-L2): means %.3g quats %.3g scales %.3g (all but the thin axes: %.3g; thin axes alone: %.3g) colours %.3g opacities %.3g" % (
        tag, rl(gs[0][fin], o_g[0][fin]), rl(gs[1][fin], o_g[1][fin]), rl(gs[2][fin], o_g[2][fin]), rl(gs[2][wide & fin[:, None]], o_g[2][wide & fin[:, None]]),
        rl(gs[2][pk, ax.numpy()], o_g[2][pk, ax.numpy()]), rl(gs[3][0][fin], o_g[3][0][fin]), rl(gs[4][0][fin], o_g[4][0][fin])))


print("float64 image vs reference kernel: %.3g" % float(np.abs(o_f[0] - r_f[0].cpu().numpy()).max()))
vs64(r_g, "reference kernel")
ut = ops.UnscentedTransformParameters()
hargs = (means, quats, scales, R["colors"], opac[None].contiguous(), bg, None, W, H, 16, vm, None, K, ops.CameraModelType.PINHOLE, ut, ops.ShutterType.GLOBAL, None, None, None, R["tile_offsets"], R["flatten_ids"])
for path, bwd in (("", ""), ("", "pm"), ("generic", "")):
    for k, v in (("GSX_RASTER_PATH", path), ("GSX_BWD", bwd)):
        if v: os.environ[k] = v
        else: os.environ.pop(k, None)
    h_f = ops.rasterize_to_pixels_from_world_3dgs_fwd(*hargs)
    h_g = ops.rasterize_to_pixels_from_world_3dgs_bwd(*hargs, h_f[1], h_f[2], v_rc, v_ra)
    rows = [int((~torch.isfinite(x.reshape(N, -1))).any(-1).sum()) for x in h_g]
    print("HIP path %-8s bwd %-3s: forward non-finite %d / %d, backward elements %s, Gaussians with a non-finite gradient %s (thin discs among them: %d)" % (
        path or "fast", bwd or "gq", nf(h_f[0]), nf(h_f[1]), [nf(x) for x in h_g], rows,
        int((~torch.isfinite(h_g[2].reshape(N, -1))).any(-1)[pick.to(DEV)].sum())))
    vs64(h_g, "HIP " + (path or "fast") + " / " + (bwd or "gq"))
    fin = torch.isfinite(h_g[2]).all(-1) & torch.isfinite(r_g[2]).all(-1)
    print("      v_scales rel-L2 on the finite rows vs the reference: %.3g; image max err %.3g" % (float((h_g[2][fin] - r_g[2][fin]).norm() / r_g[2][fin].norm()), float((h_f[0] - r_f[0]).abs().nan_to_num(9).max())))
