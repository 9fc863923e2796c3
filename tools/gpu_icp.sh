#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_tracking_gpu.py -x -q --timeout 120 --timeout-method thread > gpurun_out/icp_tests.out 2>&1
rc=$?; tail -8 gpurun_out/icp_tests.out
[ $rc -ne 0 ] && exit $rc
for i in 1 2; do
timeout -k 10 200 python tools/bench_loop.py --frames 300 | head -2 || exit 1
timeout -k 10 200 python tools/bench_loop.py --frames 300 --chain | head -2 || exit 1
done
