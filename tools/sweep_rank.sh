#!/bin/bash
# usage: tools/sweep_rank.sh   (on the GPU box): per-rank slab time of a multi-GPU job, by z-chunk
cd $GRAFT_REPO_ROOT
for rank in 8:0 8:7 4:3 1:0; do
  for zc in 0 4 8 16; do
    timeout -k 10 100 python bench.py --no-cpu-baseline --no-rows --as-rank $rank --z-chunk $zc --steps 100 | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline())
print('rank $rank z_chunk $zc kernel_ms %.4f step_ms %.4f' % (r['roofline']['kernel_ms'], r['ms_per_step']))" || exit 1
  done
done
