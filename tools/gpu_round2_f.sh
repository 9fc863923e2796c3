#!/bin/bash
set -u
mkdir -p gpurun_out
step() { local name=$1 lim=$2; shift 2
  timeout -k 10 "$lim" "$@" > "gpurun_out/$name.out" 2> "gpurun_out/$name.err"; local rc=$?
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "stopping after $name rc=$rc"; tail -5 gpurun_out/$name.out gpurun_out/$name.err; exit $rc; fi; }
step smap 900 python -u -m pytest tests/test_surface_map_gpu.py -x -q --timeout 300 --timeout-method thread
tail -15 gpurun_out/smap.out
step rest 1100 python -u -m pytest tests/test_raycast_gpu.py tests/test_slab_tracking_gpu.py tests/test_sharded_kinfu_gpu.py tests/test_integrate_gpu.py tests/test_integrate_batch_gpu.py tests/test_zz_no_alloc.py tests/test_host_cpp.py tests/test_comm_gpu.py -x -q --timeout 300 --timeout-method thread
tail -5 gpurun_out/rest.out
step bench 600 python bench.py --steps 20 --warmup 5
tail -1 gpurun_out/bench.out | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('frac',d['roofline']['frac'],'kernel_ms',d['roofline']['kernel_ms'])
print(json.dumps(d['next_rows']['raycast'])[:1500])
print(json.dumps(d['kinfu_loop'])[-400:])
print(json.dumps(d['orbit_views'])[:300])
"
