"""Housekeeping: the driver runs `pytest -m "not gpu"` on a box without a GPU — every test of a tests/test_gpu_*.py file must carry the gpu marker
(module-level `pytestmark` or a per-test decorator), or it runs there and fails."""
import ast
import glob
import os


def test_every_test_of_a_gpu_file_is_gpu_marked():
    here = os.path.dirname(os.path.abspath(__file__))
    unmarked = []
    for path in sorted(glob.glob(os.path.join(here, "test_gpu_*.py"))):
        tree = ast.parse(open(path).read())
        if any(isinstance(n, ast.Assign) and any(getattr(t, "id", "") == "pytestmark" for t in n.targets) for n in tree.body):
            continue
        for n in tree.body:
            if isinstance(n, ast.FunctionDef) and n.name.startswith("test_") and not any("mark.gpu" in ast.unparse(d) for d in n.decorator_list):
                unmarked.append("%s::%s" % (os.path.basename(path), n.name))
    assert not unmarked, unmarked
