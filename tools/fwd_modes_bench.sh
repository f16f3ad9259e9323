export GSX_TEST_SWITCHES=1
for r in 1 2; do for m in quad pair default; do
  if [ $m = default ]; then unset GSX_FWD; else export GSX_FWD=$m; fi
  python bench.py --no-cpu-baseline --no-order-ablation --no-camera-batch --no-s5m --no-fwd-bwd --steps 30 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('$m step %.4f fwd %.4f bwd %.4f' % (d['ms_per_step'], k['rasterize_to_pixels_from_world_3dgs_fwd']['ms'], k['rasterize_to_pixels_from_world_3dgs_bwd']['ms']))"
done; done
