#!/usr/bin/env bash
# short runs of examples/train_colmap.py on the synthetic capture until one aborts: bash tools/soak_colmap.sh <runs> <iters> [ENV=VALUE ...]   (python -X faulthandler: the Python stack of an abort)
cd "$GRAFT_REPO_ROOT"
runs=$1; iters=$2; shift 2
[ -d /tmp/syn ] || python examples/train_colmap.py --make-synthetic /tmp/syn >/dev/null 2>&1
ok=0; bad=0; net=0
for i in $(seq $runs); do
  env "$@" timeout 600 python -u -X faulthandler examples/train_colmap.py -d /tmp/syn --iter $iters --log-every 500 -o /tmp/syn_out > /tmp/soak_c.log 2>&1
  net=$((net + $(grep -c "hold non-finite parameters" /tmp/soak_c.log)))
  if grep -q '"iterations"' /tmp/soak_c.log; then ok=$((ok+1)); else bad=$((bad+1)); echo "--- run $i failed:"; grep -v "amdgpu.ids" /tmp/soak_c.log | grep -v "^  File \"/usr" | tail -40 | cut -c1-220; fi
done
echo "config [$*] x $runs runs of $iters iterations: ok $ok, failed $bad, refine events that relocated non-finite Gaussians $net"
