cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05g; mkdir -p $out
bash tools/pmc_passes.sh "$out/pmc" > "$out/pmc.log" 2>&1; tail -2 "$out/pmc.log"
python tools/pmc_to_json.py "$out/pmc/summary.txt" s1m_1080p "profiles/r05f_pmc_counters.md (rocprofv3 --pmc, separate passes, tools/pmc_passes.sh; bench step with the cfg2 camera)" 3360791 > "$out/pmc_entry.json" && cp profiles/pmc.json "$out/pmc.json"
python bench.py > "$out/bench.json" 2> "$out/bench.err"; echo bench rc=$?
