#!/bin/bash
# usage: bash tools/gpu_ab.sh   -- S0/S1/S2 kernel times of the current build (bench.py, 100 steps, integrate only)
set -u
mkdir -p gpurun_out
for sc in s0 s1 s2; do
  timeout -k 10 120 python bench.py --steps 100 --warmup 10 --no-rows --no-cpu-baseline --no-loop --scene $sc > gpurun_out/ab_$sc.out 2> gpurun_out/ab_$sc.err || exit 1
  tail -1 gpurun_out/ab_$sc.out | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$sc kernel_ms', r['kernel_ms'], 'steady', r['steady_state']['kernel_ms'], 'frac', r['frac'])"
done
