"""Times one Adam step over the 59 floats of 1 M Gaussians through the single-tensor operator (six groups, as fused_adam.cpp walks them):
the A/B harness of the optimizer kernels' cache policy (nontemporal loads / stores: 0.299 -> 0.276 ms).   python tools/adam_ab.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import gsx  # noqa: E402,F401
from gsx import ops  # noqa: E402

N=1_000_000
groups=[torch.randn(N*k, device="cuda") for k in (3,3,45,3,4,1)]
state=[(torch.zeros_like(g), torch.zeros_like(g), torch.randn_like(g)) for g in groups]
def run():
    for p,(m,v,g) in zip(groups,state): ops.adam_step(p,m,v,g,1e-3,0.9,0.999,1e-15,10.0,31.6)
for _ in range(3): run()
torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print("adam 59 floats x 1M: %.4f ms" % (e0.elapsed_time(e1)/20))
