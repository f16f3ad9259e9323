"""rel-L2 between the blend-backward gradients two kernel variants saved with GSX_AB_SAVE (tools/blend_ab.py).
python tools/blend_ab_compare.py ref.pt other.pt [other2.pt ...]"""
import sys

import torch

names = ["v_means", "v_quats", "v_scales", "v_colors", "v_opacities"]
ref = torch.load(sys.argv[1])
for p in sys.argv[2:]:
    oth = torch.load(p)
    out = []
    for n, a, b in zip(names, ref, oth):
        a, b = a.double(), b.double()
        out.append("%s %.2e" % (n, float((a - b).norm() / a.norm().clamp_min(1e-300))))
    print(p, "vs", sys.argv[1], "rel-L2:", "  ".join(out))
