#!/usr/bin/env bash
# repeat tools/soak_run.py <iters> under a configuration and count the outcomes: bash tools/soak_repeat.sh <runs> <iters> [ENV=VALUE ...]
cd "$GRAFT_REPO_ROOT"
runs=$1; iters=$2; shift 2
ok=0; nan=0; crash=0
for i in $(seq $runs); do
  out=$(env GSX_TEST_SWITCHES=1 "$@" timeout 300 python tools/soak_run.py $iters 2>&1 | grep "^soak:\|HSA_STATUS\|Error\|error" | tail -2)
  if echo "$out" | grep -q "soak: ok"; then ok=$((ok+1));
  elif echo "$out" | grep -q "soak: non-finite"; then nan=$((nan+1)); echo "   run $i: $out";
  else crash=$((crash+1)); echo "   run $i: $(echo "$out" | tail -1 | cut -c1-160)"; fi
done
echo "config [$*] x $runs runs of $iters iterations: ok $ok, non-finite $nan, crashed $crash"
