#!/bin/bash
set -u
mkdir -p gpurun_out; rm -f gpurun_out/b_*.out gpurun_out/b_*.err
step() { local name=$1 lim=$2; shift 2
  timeout -k 10 "$lim" "$@" > "gpurun_out/$name.out" 2> "gpurun_out/$name.err"; local rc=$?
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "stopping after $name rc=$rc"; exit $rc; fi; }
step parity 600 python -u -m pytest tests/test_integrate_gpu.py tests/test_integrate_batch_gpu.py tests/test_zz_no_alloc.py -x -q --timeout 300 --timeout-method thread
tail -3 gpurun_out/parity.out
B="python bench.py --steps 100 --warmup 10 --no-rows --no-cpu-baseline"
for sc in s0 s1 s2; do
  for t in 3 4 5 6; do step b_${sc}_side_t$t 120 $B --scene $sc --tile $t; done
  step b_${sc}_side_t4_c16 120 $B --scene $sc --tile 4 --z-chunk 16
  step b_${sc}_side_t3_c16 120 $B --scene $sc --tile 3 --z-chunk 16
done
for f in gpurun_out/b_*.out; do python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    print(sys.argv[1].split('/')[-1], 'kernel_ms', r['kernel_ms'], 'steady', r['steady_state']['kernel_ms'], 'step', r['step_ms_device'], 'value', d['value'])
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
