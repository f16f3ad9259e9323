#!/usr/bin/env bash
# GPU-box helper: parity tests, then the bench summary (per-op ms + HBM fraction).  Usage: bash tools/gpu_check.sh [pytest -k expr]
mkdir -p gpurun_out
if [ -n "$1" ]; then python -m pytest tests -m gpu -q -x -k "$1" 2>&1 | tail -8; else python -m pytest tests -m gpu -q -x 2>&1 | tail -8; fi
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>gpurun_out/bench.err | tee gpurun_out/bench_last.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms_per_step', d['ms_per_step'], ' frames/s', d['value'])
for k,v in d['kernels'].items(): print('  %-45s %8.4f ms  %6.1f GB/s  %.3f of HBM' % (k, v['ms'], v['GBps'], v['frac_hbm']))
"
