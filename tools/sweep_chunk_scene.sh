#!/bin/bash
# usage: tools/sweep_chunk_scene.sh [z_chunk ...]  (on the GPU box): integrate kernel time at 512^3 by scene and z-chunk (0 = auto)
cd $GRAFT_REPO_ROOT
for sc in s0 s1 s2; do
  for zc in ${@:-0}; do
    timeout -k 10 100 python bench.py --no-cpu-baseline --no-rows --scene $sc --z-chunk $zc --steps 100 | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline())
print('scene $sc z_chunk $zc kernel_ms %.4f step_ms %.4f' % (r['roofline']['kernel_ms'], r['ms_per_step']))" || exit 1
  done
done
