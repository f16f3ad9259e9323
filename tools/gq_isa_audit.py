"""ISA audit of the Gaussian-major backward's pass loop (profiles/r05_gq_isa.md): classifies and counts the instructions of ONE pass of
raster_bwd_gq_kernel<pinhole> (the arm without the alpha clamp) in hipcc's assembly and prices them with the issue costs measured by
tools/valu_clock_probe.hip / tools/valu_pattern_probe.hip.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -S --cuda-device-only gaussian-splatting-cuda_amd/csrc/gsx_raster_fast.hip -o /tmp/rf.s
    python tools/gq_isa_audit.py /tmp/rf.s"""
import collections
import re
import sys


def classify(op, rest):
    if op.startswith(("v_exp", "v_rcp", "v_permlane", "v_rsq", "v_log")):
        return "quarter-rate (8.2)"
    if "row_shr" in rest or "quad_perm" in rest or "row_" in rest or op.endswith("_dpp"):
        return "DPP (4.1 in runs)"
    if op.startswith(("v_mad_u32_u24", "v_readlane", "v_readfirstlane", "v_pk_")):
        return "slow-pairing (4.1 - 8.7)"
    if op.startswith("v_"):
        return "plain VALU (2.3)"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith("s_"):
        return "SALU / wait / branch"
    return "other"


def main(path):
    lines = open(path).read().split("\n")
    k0 = next(i for i, l in enumerate(lines) if l.startswith("_ZN3gsx20raster_bwd_gq_kernelILi0E"))
    k1 = next(i for i in range(k0, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[k0:k1]
    exps = [i for i, l in enumerate(body) if "v_exp_f32" in l]
    first_exp = exps[0]
    header = max(i for i in range(first_exp) if re.match(r"\.LBB\d+_\d+:", body[i]) and "Depth=3" in " ".join(body[i:i + 3]))
    # walk ONE pass along the clamp-free arm: take the clamp branch (s_cbranch_vccz: "no lane can reach alpha 0.999"), follow unconditional
    # branches, fall through exec-mask skips (their bodies run for some lanes) and through the lock's spin loop; stop back at the header
    labels = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"\.LBB\d+_\d+:", l)}
    seq, i, seen, took_clamp, target = [], header, set(), False, "?"
    while i not in seen and i < len(body):
        seen.add(i)
        l = body[i].strip()
        seq.append(body[i])
        op = l.split()[0] if l else ""
        if op == "s_cbranch_vccz" and not took_clamp and i > first_exp - 200:
            took_clamp, target = True, l.split()[-1]
            i = labels[target]
            continue
        if op == "s_branch":
            i = labels[l.split()[-1]]
            if i == header:
                break
            continue
        if op.startswith("s_cbranch") and labels.get(l.split()[-1], -1) == header:
            break
        i += 1
    clamp_lines = 0
    cls, ops = collections.Counter(), collections.Counter()
    for l in seq:
        l = l.strip()
        if not l or l.startswith((";", ".")) or l.endswith(":"):
            continue
        m = re.match(r"(\S+)\s*(.*)", l)
        c = classify(m.group(1), m.group(2))
        cls[c] += 1
        ops[(c, re.sub(r"_e32|_e64", "", m.group(1)))] += 1
    valu = sum(v for k, v in cls.items() if k not in ("LDS", "SALU / wait / branch", "other"))
    cyc = 2.3 * cls["plain VALU (2.3)"] + 8.2 * cls["quarter-rate (8.2)"] + 4.1 * cls["DPP (4.1 in runs)"] + 6.0 * cls["slow-pairing (4.1 - 8.7)"]
    print("pass loop %s .. %s (clamp-free arm): %d instructions, %d VALU, ~%.0f issue cycles" % (body[header].split(":")[0], target, sum(cls.values()), valu, cyc))
    print(dict(cls))
    for (c, o), n in sorted(ops.items(), key=lambda kv: (kv[0][0], -kv[1])):
        print("%-26s %-26s %d" % (c, o, n))


if __name__ == "__main__":
    main(sys.argv[1])
