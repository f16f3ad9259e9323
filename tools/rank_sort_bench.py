"""Times gsx_intersect_depth_ranks (raw C ABI) at C*N = 1 M: the hand-written radix sort vs the library sort (GSX_RANK_SORT=rocprim).
Usage: GSX_TEST_SWITCHES=1 python tools/rank_sort_bench.py [N]"""
import ctypes
import os
import sys

import numpy as np
import torch

os.environ.setdefault("GSX_TEST_SWITCHES", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "gaussian-splatting-cuda_amd", "libgsx.so"))
lib.gsx_intersect_depth_ranks_workspace_bytes.restype = ctypes.c_size_t
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
dev = "cuda:0"
rng = np.random.default_rng(0)
D = torch.from_numpy(rng.uniform(0.5, 40.0, N).astype(np.float32)).to(dev)
R = torch.from_numpy(rng.integers(0, 8, (N, 2)).astype(np.int32)).to(dev)
ranks = torch.empty(N, dtype=torch.int32, device=dev)
order = torch.empty(N, dtype=torch.int32, device=dev)
wb = lib.gsx_intersect_depth_ranks_workspace_bytes(ctypes.c_uint32(1), ctypes.c_uint32(N))
ws = torch.empty(wb, dtype=torch.uint8, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())


def run():
    rc = lib.gsx_intersect_depth_ranks(ctypes.c_uint32(1), ctypes.c_uint32(N), p(R), p(D), p(ranks), p(order), p(ws), ctypes.c_size_t(wb), None)
    assert rc == 0


for mode in ("own", "rocprim", "own", "rocprim"):
    if mode == "rocprim":
        os.environ["GSX_RANK_SORT"] = "rocprim"
    else:
        os.environ.pop("GSX_RANK_SORT", None)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run()
    e1.record()
    torch.cuda.synchronize()
    print(f"{mode:8s} N={N}: {e0.elapsed_time(e1) / 50 * 1000:.1f} us per call")
