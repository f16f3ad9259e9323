"""Generates tests/golden/colmap_ref/<case>.txt: what the REFERENCE's own COLMAP reader (src/loader/formats/colmap.cpp compiled unmodified
into oracle/_ref/colmap_ref_tool by oracle/build_ref_colmap.sh) parses from the deterministic models of tests/golden/colmap_cases.py.
Run in the build container (needs /root/reference):  python tests/golden/gen_colmap_ref_golden.py"""
import os
import pathlib
import subprocess
import sys
import tempfile

HERE = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import colmap_cases  # noqa: E402

TOOL = HERE.parent.parent / "oracle" / "_ref" / "colmap_ref_tool"


def run_tool(root, folder, text):
    r = subprocess.run([str(TOOL), str(root), folder, "text" if text else "bin"], capture_output=True, text=True)
    return r.stdout


if __name__ == "__main__":
    assert TOOL.exists(), "build it first: bash oracle/build_ref_colmap.sh"
    out = HERE / "colmap_ref"
    out.mkdir(exist_ok=True)
    for name, kw in colmap_cases.CASES.items():
        with tempfile.TemporaryDirectory() as d:
            root = pathlib.Path(d)
            colmap_cases.write_model(root, **kw)
            txt = run_tool(root, kw.get("images_folder", "images"), kw.get("text", False))
        (out / f"{name}.txt").write_text(txt)
        print(name, len(txt.splitlines()), "lines", os.path.getsize(out / f"{name}.txt"), "bytes")
