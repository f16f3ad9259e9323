"""Reader / writer of the raw tensor directories exchanged with tools/dump_reference_cuda.cpp: <dir>/<name>.bin (little-endian,
C order) + <dir>/manifest.txt with one line per tensor "name dtype ndim d0 d1 ..." (dtype in f32 i32 i64 u8)."""
import os

import numpy as np

_DT = {"f32": np.float32, "i32": np.int32, "i64": np.int64, "u8": np.uint8}
_NAME = {np.dtype(v): k for k, v in _DT.items()}


def write_dir(path, tensors):
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "manifest.txt"), "w") as m:
        for name, a in tensors.items():
            a = np.ascontiguousarray(a)
            if a.dtype == np.bool_:
                a = a.astype(np.uint8)
            a.astype(a.dtype.newbyteorder("<")).tofile(os.path.join(path, name + ".bin"))
            m.write("%s %s %d %s\n" % (name, _NAME[a.dtype], a.ndim, " ".join(str(d) for d in a.shape)))


def read_dir(path):
    out = {}
    with open(os.path.join(path, "manifest.txt")) as m:
        for line in m:
            f = line.split()
            if len(f) < 3:
                continue
            name, dt, nd = f[0], _DT[f[1]], int(f[2])
            shape = tuple(int(x) for x in f[3:3 + nd])
            a = np.fromfile(os.path.join(path, name + ".bin"), dtype=np.dtype(dt).newbyteorder("<"))
            out[name] = a.reshape(shape)
    return out
