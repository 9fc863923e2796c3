#!/bin/bash
# usage: tools/sweep_rank.sh [z_chunk ...]  (on the GPU box): per-rank slab time of a multi-GPU job, by z-chunk (0 = auto)
cd $GRAFT_REPO_ROOT
for rank in 1:0 2:1 4:0 4:3 8:0 8:7; do
  for zc in ${@:-0}; do
    timeout -k 10 100 python bench.py --no-cpu-baseline --no-rows --as-rank $rank --z-chunk $zc --steps 100 | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline())
print('rank $rank z_chunk $zc kernel_ms %.4f step_ms %.4f' % (r['roofline']['kernel_ms'], r['ms_per_step']))" || exit 1
  done
done
