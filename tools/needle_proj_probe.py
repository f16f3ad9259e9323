"""UT projection of needle-shaped Gaussians: HIP, the reference kernel and the fp32 oracle against the float64 oracle (conics, radii).
GPU box: python tools/needle_proj_probe.py [regime]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import gsx  # noqa: E402,F401
from gsx import ops, scenes  # noqa: E402
from oracle import oracle, ref_hip  # noqa: E402
import tests.test_gpu_reference_hip as T  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "needles"
sc = T._regime(scenes, name)
ref = ref_hip.load()
a = T._scene_args(sc, {})
W, H = a["width"], a["height"]
cm, shut = T._hip_enums(ops, a)
ut = ops.UnscentedTransformParameters()
P = ops.projection_ut_3dgs_fused(a["means"], a["quats"], a["scales"], a["opacities"], a["viewmat"], None, a["K"], W, H, 0.3, 0.01, 1e4, 0.0, False, cm, ut, shut, None, None, None)
R = ref.projection_ut_3dgs_fused(a["means"], a["quats"], a["scales"], a["opacities"], a["viewmat"], None, a["K"], W, H, 0.3, 0.01, 1e4, 0.0, False, ref_hip.PINHOLE, None, ref_hip.GLOBAL, None, None, None)
f = lambda k, dt: np.ascontiguousarray(sc[k].numpy(), dt)  # noqa: E731
o32 = oracle.projection_ut(f("means", np.float32), f("quats", np.float32), f("scales", np.float32), f("opacities", np.float32), f("viewmat", np.float32)[None], f("K", np.float32)[None], W, H)
o64 = oracle.projection_ut(f("means", np.float64), f("quats", np.float64), f("scales", np.float64), f("opacities", np.float64), f("viewmat", np.float64)[None], f("K", np.float64)[None], W, H)
vis = (o64[0] > 0).all(-1)[0]
def report(who, radii, means2d, conics):
    radii, means2d, conics = np.asarray(radii)[0], np.asarray(means2d, np.float64)[0], np.asarray(conics, np.float64)[0]
    v = vis & (radii > 0).all(-1)
    rel = np.abs(conics[v] - o64[3][0][v]).max(-1) / np.abs(o64[3][0][v]).max(-1)
    print("%-10s vs float64: radius flips %4d (max %d px)  conic rel err max %.2e median %.2e  means2d max err %.2e px" % (
        who, int((radii[v] != o64[0][0][v]).any(-1).sum()), int(np.abs(radii[v] - o64[0][0][v]).max()), rel.max(), np.median(rel), np.abs(means2d[v] - o64[1][0][v]).max()))
report("HIP", P[0].cpu().numpy(), P[1].cpu().numpy(), P[3].cpu().numpy())
if R is not None:
    report("reference", R[0].cpu().numpy(), R[1].cpu().numpy(), R[3].cpu().numpy())
report("oracle f32", o32[0], o32[1], o32[3])
