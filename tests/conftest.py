import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("GSX_TEST_SWITCHES", "1")   # the tests flip libgsx's A/B switches (GSX_RASTER_PATH, GSX_BWD, ...): include/gsx.h gsx_test_switch
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    # the extension is built in-tree by __graft_entry__.build(); build it here if a fresh checkout has not done so yet
    lib = os.path.join(ROOT, "gaussian-splatting-cuda_amd", "libgsx.so")
    if not os.path.exists(lib):
        import subprocess
        subprocess.check_call([sys.executable, os.path.join(ROOT, "gaussian-splatting-cuda_amd", "build.py")])


def pytest_collection_modifyitems(config, items):
    # GPU tests are only collected for execution when a GPU is present; on CPU they are skipped so
    # `-m "not gpu"` and a plain run both stay green.
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


GOLDEN = os.path.join(ROOT, "tests", "golden")
