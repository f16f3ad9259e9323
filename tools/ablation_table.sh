#!/usr/bin/env bash
# One box, one table: what each piece of the fused training iteration buys at S-1M (bench.py's own switches, three 20-step regions each, median):
#   bash tools/ablation_table.sh > gpurun_out/r06/ablation.md      (GPU box)
cd "$GRAFT_REPO_ROOT"
common="--steps 20 --warmup 5 --no-cpu-baseline --no-camera-batch --no-s5m --no-order-ablation --sustained-steps 0"
echo "| configuration | ms per iteration (3 regions) | it/s | fwd+bwd ms/frame |"
echo "|---|---|---|---|"
run() {
  name=$1; shift
  python bench.py $common "$@" > /tmp/abl.json 2> /tmp/abl.err || { echo "| $name | failed: $(tail -1 /tmp/abl.err | cut -c1-120) | | |"; return; }
  python - "$name" <<'PY'
import json, sys
d = json.load(open("/tmp/abl.json"))
r = d.get("repeats", {}).get("ms_per_step_each", [d["ms_per_step"]])
print("| %s | %s | %.1f | %s |" % (sys.argv[1], " / ".join("%.4f" % x for x in r), d["value"], d.get("fwd_bwd", {}).get("ms_per_frame", "-")))
PY
}
run "default (fused front end, guarded lists, act epilogue, SH Adam in the SH backward, Morton order)"
run "\`--exact-lists\` (the reference's protocol: the host reads n_isects inside intersect_tile)" --exact-lists
run "\`--unfused-adam\` (SH gradient written, SH groups stepped by the separate Adam launch)" --unfused-adam
run "\`--unfused\` (reference-style glue: one torch op per activation / SH pre- and post-step, plain Ops.h operators)" --unfused
run "\`--random-order\` (Gaussians stored as generated)" --random-order
run "\`--fixed-camera\` (the cfg2 camera on every step)" --fixed-camera
run "\`--l1-loss\` (plain torch L1 instead of the fused L1 + SSIM)" --l1-loss
run "default again (drift of the box over the table)"
