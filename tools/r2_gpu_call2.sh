#!/usr/bin/env bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
: > gpurun_out/ab2.log
for v in gm pm; do GSX_BWD=$v timeout 300 python tools/blend_ab.py 1m 20 2>&1 | tail -1 >> gpurun_out/ab2.log; done
for v in pm gm; do GSX_BWD=$v timeout 300 python tools/blend_ab.py 5m 10 2>&1 | tail -1 >> gpurun_out/ab2.log; done
GSX_AB_CAMERA=fisheye timeout 300 python tools/blend_ab.py 1m 10 2>&1 | tail -1 >> gpurun_out/ab2.log
cat gpurun_out/ab2.log
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_cameras.py tests/test_gpu_fused.py tests/test_gpu_edge_cases.py tests/test_gpu_fullsize.py -q -m gpu -x -k "not s5m" > gpurun_out/t2.log 2>&1
tail -5 gpurun_out/t2.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench1.log 2> gpurun_out/bench1.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench1.log').read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'fwd_bwd', d.get('fwd_bwd'))
for k,v in d['kernels'].items(): print('  %-45s %8.4f ms  %6.1f GB/s' % (k, v['ms'], v['GBps']))
PY
