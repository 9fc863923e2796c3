#!/bin/bash
# the tracking tests, then the loop three times (first two lines of each)
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_tracking_gpu.py -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/icp_tests.out 2>&1 || { tail -30 gpurun_out/icp_tests.out; exit 1; }
tail -1 gpurun_out/icp_tests.out
for i in 1 2 3; do
  timeout -k 10 200 python tools/bench_loop.py --frames 300 > gpurun_out/loop_ab.out 2> gpurun_out/loop_ab.err || { tail -3 gpurun_out/loop_ab.err; exit 1; }
  sed -n 1,2p gpurun_out/loop_ab.out
done
