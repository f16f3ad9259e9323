"""Where do the scale gradients of needle-shaped Gaussians differ?  HIP blend backward (both kernels) vs the reference's kernel vs the
float64 oracle on the `needles` regime of tests/test_gpu_reference_hip.py.  GPU box: python tools/needle_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import gsx  # noqa: E402,F401
from gsx import ops, scenes  # noqa: E402
from oracle import oracle, ref_hip  # noqa: E402
import tests.test_gpu_reference_hip as T  # noqa: E402
from tests.helpers import np32, rel_l2  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "needles"
sc = T._regime(scenes, name)
ref = ref_hip.load()
a = T._scene_args(sc, {})
v_rc, v_ra = T._grads(sc)
W, H = a["width"], a["height"]
R = ref_hip.render_chain(ref, a["means"], a["quats"], a["scales"], a["opacities"], a["sh"], a["sh_degree"], a["viewmat"], a["K"], W, H, a["background"],
                         v_render_colors=v_rc, v_render_alphas=v_ra, camera_model=0, shutter=ref_hip.GLOBAL, viewmats1=None, radial=None, tangential=None,
                         thin_prism=None, calc_compensations=False)
cm, shut = T._hip_enums(ops, a)
ut = ops.UnscentedTransformParameters()
common = (a["means"], a["quats"], a["scales"], R["colors"], a["opacities"][None].contiguous(), a["background"], None, W, H, 16, a["viewmat"], None, a["K"], cm, ut, shut,
          None, None, None, R["tile_offsets"], R["flatten_ids"])
f64 = lambda k: np.ascontiguousarray(sc[k].numpy(), np.float64)  # noqa: E731
d64 = lambda t: t.detach().cpu().numpy().astype(np.float64)  # noqa: E731
o64 = oracle.rasterize_bwd(f64("means"), f64("quats"), f64("scales"), d64(R["colors"]), f64("opacities")[None], f64("background")[None], None, W, H, 16,
                           f64("viewmat")[None], f64("K")[None], R["tile_offsets"].cpu().numpy(), R["flatten_ids"].cpu().numpy(), d64(R["alphas"]),
                           R["last_ids"].cpu().numpy(), d64(v_rc), d64(v_ra))
names = ["v_means", "v_quats", "v_scales", "v_colors", "v_opacities"]
res = {"reference": [np32(R[n]) for n in names]}
for path, bwd in (("fast", "pm"), ("fast", "gq"), ("generic", "gq")):
    os.environ["GSX_BWD"] = bwd
    if path == "generic":
        os.environ["GSX_RASTER_PATH"] = "generic"
    B = ops.rasterize_to_pixels_from_world_3dgs_bwd(*common, R["alphas"], R["last_ids"], v_rc, v_ra)
    os.environ.pop("GSX_RASTER_PATH", None)
    res["hip %s/%s" % (path, bwd)] = [np32(g) for g in B]
for who, g in res.items():
    print("%-18s rel-L2 vs float64: " % who + "  ".join("%s %.2e" % (n, rel_l2(x.astype(np.float64), o)) for n, x, o in zip(names, g, o64)))
s = sc["scales"].numpy()
order = np.argsort(-s, axis=1)   # longest axis first
for who, g in res.items():
    vs, o = g[2].reshape(-1, 3).astype(np.float64), o64[2].reshape(-1, 3)
    parts = []
    for rank, label in enumerate(("long axis", "middle", "short")):
        idx = order[:, rank]
        x, y = vs[np.arange(len(vs)), idx], o[np.arange(len(vs)), idx]
        parts.append("%s: rel %.2e (|ref| %.2e)" % (label, np.linalg.norm(x - y) / np.linalg.norm(y), np.linalg.norm(y)))
    print("%-18s v_scales by axis length vs float64:  " % who + "   ".join(parts))
hip = res["hip fast/gq"][2].reshape(-1, 3).astype(np.float64)
o = o64[2].reshape(-1, 3)
err = np.abs(hip - o).max(1)
worst = np.argsort(-err)[:8]
for i in worst:
    print("gaussian %4d scales %s  v_scales f64 %s  hip %s  reference %s  z %.2f" % (i, s[i], o[i], hip[i], res["reference"][2].reshape(-1, 3)[i], sc["means"][i, 2]))
