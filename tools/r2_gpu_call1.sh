#!/usr/bin/env bash
# round-2 session-2 GPU call 1: MFMA probe, backward variant A/B, parity subset, 2-rank functional bench, 1-GPU bench
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o /tmp/mfma_probe 2>/dev/null && timeout 120 /tmp/mfma_probe > gpurun_out/mfma_probe.log 2>&1
: > gpurun_out/ab.log
for v in gmv gm gm3 pm; do GSX_BWD=$v GSX_AB_SAVE=/tmp/ab_$v.pt timeout 300 python tools/blend_ab.py 1m 20 2>&1 | tail -2 >> gpurun_out/ab.log; done
timeout 120 python tools/blend_ab_compare.py /tmp/ab_gmv.pt /tmp/ab_gm.pt /tmp/ab_gm3.pt /tmp/ab_pm.pt >> gpurun_out/ab.log 2>&1
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fused.py tests/test_gpu_distributed.py tests/test_gpu_fullsize.py -q -m gpu -k "not s5m" > gpurun_out/t1.log 2>&1
tail -15 gpurun_out/t1.log
GSX_BENCH_ALL_RANKS_ON_DEVICE0=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --no-cpu-baseline --no-fwd-bwd > gpurun_out/bench2.log 2>&1
tail -3 gpurun_out/bench2.log | cut -c1-1500
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench1.log 2> gpurun_out/bench1.err
cat gpurun_out/mfma_probe.log gpurun_out/ab.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench1.log').read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], 'value', d['value'], 'fwd_bwd', d.get('fwd_bwd'))
for k,v in d['kernels'].items(): print('  %-45s %8.4f ms  %6.1f GB/s' % (k, v['ms'], v['GBps']))
PY
