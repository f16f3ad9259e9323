"""rocprofv3 --pmc summary (tools/pmc_summary.py output) -> profiles/pmc.json entry for one workload: per blend op the HBM traffic
((2 * FETCH_SIZE + WRITE_SIZE) KiB, the gfx950 correction of MI355X_MICROARCH.md, summed over the kernels of the op) and the VALU
instruction count of its main kernel.  bench.py reads it for `roofline.traffic` and `roofline_valu` — only for the workload it was
measured on.   python tools/pmc_to_json.py gpurun_out/r02/pmc/summary.txt s1m_1080p "profiles/r02_pmc_counters.md" """
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse(path):
    kernels, cur = {}, None
    for line in open(path):
        if not line.startswith(" "):
            cur = line.strip()
            kernels[cur] = {}
        else:
            m = re.match(r"\s+(\w+)\s+([0-9.]+) per dispatch", line)
            if m:
                kernels[cur][m.group(1)] = float(m.group(2))
    return kernels


def main():
    summary, key, source = sys.argv[1], sys.argv[2], sys.argv[3]
    k = parse(summary)
    find = lambda sub: next((v for n, v in k.items() if sub in n), None)  # noqa: E731
    traffic = lambda v: int((2 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024) if v else 0  # noqa: E731
    pack, gather = find("pack_records"), find("gsx_bwd_gather")
    fwd_name = next((n for n in ("raster_fwd_pair", "raster_fwd_quad") if find(n)), "raster_fwd_fast")   # whichever forward kernel the launcher took for this workload
    fwd = find(fwd_name)
    bwd = find("raster_bwd_gq") or find("raster_bwd_gm") or find("raster_bwd_fast")
    sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-cuda_amd"))
    import build as gbuild
    entry = {
        "source": source,
        "blend_kernel_hash": gbuild.blend_kernel_hash(),   # bench.py reports these figures only while the blend sources still hash to this
        # tile intersections of the frame the counters were collected on (tools/run_fwd_bwd.py renders the scene's own camera): bench.py prices
        # the counter traffic against the algorithmic bytes AT THIS COUNT, not at the mean of its timed steps' other cameras
        "n_isects": int(sys.argv[4]) if len(sys.argv) > 4 else None,
        "rasterize_to_pixels_from_world_3dgs_fwd": {"hbm_bytes": traffic(pack) + traffic(fwd), "valu_insts": int(fwd["SQ_INSTS_VALU"]),
                                                    "kernels": ("pack_records + " + fwd_name if pack else fwd_name + " (records packed by the fused front end)")},
        "rasterize_to_pixels_from_world_3dgs_bwd": {"hbm_bytes": traffic(bwd) + traffic(gather), "valu_insts": int(bwd["SQ_INSTS_VALU"]),
                                                    "kernels": "raster_bwd_gq (or raster_bwd_fast) + gsx_bwd_gather; packed records reused from the forward"},
    }
    path = os.path.join(ROOT, "profiles", "pmc.json")
    data = json.load(open(path)) if os.path.exists(path) else {}
    data[key] = entry
    json.dump(data, open(path, "w"), indent=1)
    print(json.dumps(entry, indent=1))


if __name__ == "__main__":
    main()
