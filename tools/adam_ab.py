import os, sys, torch
sys.path.insert(0, "/root/repo"); 
ROOT=os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import importlib
gsx = importlib.import_module("gaussian-splatting-cuda_amd"); sys.modules.setdefault("gsx", gsx)
from gsx import ops
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
