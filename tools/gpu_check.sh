#!/bin/bash
# usage (on the GPU box): bash tools/gpu_check.sh <pytest args...>   -- a test run under a time limit, then the driver's bench line
set -u
mkdir -p gpurun_out
timeout -k 10 1000 python -u -m pytest "$@" -x -q --timeout 300 --timeout-method thread > gpurun_out/check_tests.out 2>&1
rc=$?; tail -6 gpurun_out/check_tests.out
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/check_bench.out 2> gpurun_out/check_bench.err || exit $?
tail -1 gpurun_out/check_bench.out | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value',d['value'],'frac',d['roofline']['frac'],'kernel_ms',d['roofline']['kernel_ms'])
print('raycast',json.dumps(d['next_rows']['raycast'])[:420])
print('loop',d['kinfu_loop']['fused_frames_per_s'],d['kinfu_loop']['fused_frames_per_s_supplied_poses'],d['kinfu_loop']['stage_ms'])
print('orbit',d['orbit_views']['kernel_ms_mean'], 'batch', d['next_rows']['integrate_batch_k4_s2_ms_per_frame'], 'mc', d['next_rows']['marching_cubes']['ms_both_passes'])
"
