"""Generator of tests/golden/ref_hip/<case>.npz: the outputs of the REFERENCE's own kernels (gsplat/*.cu compiled unmodified for gfx950,
oracle/build_ref_hip.sh) for the cases of tests/golden/ref_hip_cases.py, executed on an MI355X:

    gpurun -- 'python tests/golden/gen_ref_hip_golden.py gpurun_out/ref_hip_golden'   # then copy the .npz files to tests/golden/ref_hip/

Only outputs are stored (inputs are regenerated from the seeds).  The fixtures pin the CPU oracle in the -m "not gpu" suite
(tests/test_oracle_ref_hip_golden.py): projection_ut_3dgs_fused, intersect_tile / intersect_offset, blend forward and blend backward."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import ref_hip  # noqa: E402
from tests.golden import ref_hip_cases  # noqa: E402


def main(out_dir):
    import gsx  # noqa: F401
    from gsx import scenes
    ref = ref_hip.load()
    assert ref is not None, "oracle/_ref/gsplat_ref_hip.so is not built"
    os.makedirs(out_dir, exist_ok=True)
    dev = lambda a: None if a is None else (a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))).to("cuda:0").contiguous()  # noqa: E731
    for name, (sc, cam) in ref_hip_cases.cases(scenes).items():
        v_rc, v_ra = ref_hip_cases.upstream_grads(sc)
        kw = {k: dev(None if cam.get(k) is None else np.asarray(cam[k], np.float32)) for k in ("viewmats1", "radial", "tangential", "thin_prism")}
        R = ref_hip.render_chain(ref, dev(sc["means"]), dev(sc["quats"]), dev(sc["scales"]), dev(sc["opacities"]), dev(sc["sh"]), sc["sh_degree"],
                                 dev(sc["viewmat"][None]), dev(sc["K"][None]), sc["width"], sc["height"], dev(sc["background"][None]),
                                 camera_model=cam.get("camera_model", ref_hip.PINHOLE), shutter=cam.get("shutter", ref_hip.GLOBAL),
                                 calc_compensations=cam.get("calc_compensations", False), v_render_colors=dev(v_rc), v_render_alphas=dev(v_ra), **kw)
        torch.cuda.synchronize()
        out = {k: R[k].cpu().numpy() for k in ref_hip_cases.GOLDEN_KEYS if R.get(k) is not None and R[k].numel() > 0}
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), **out)
        print(name, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "ref_hip_golden"))
