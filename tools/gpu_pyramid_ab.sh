#!/bin/bash
# yakhip_raycast_points_pyramid: its tests, the tests of its callers, then the loop three times
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_raycast_pyramid_gpu.py tests/test_raycast_gpu.py tests/test_surface_map_gpu.py tests/test_tracking_gpu.py tests/test_host_cpp.py -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/pyr_tests.out 2>&1 || { tail -40 gpurun_out/pyr_tests.out; exit 1; }
tail -2 gpurun_out/pyr_tests.out
for i in 1 2 3; do
  timeout -k 10 200 python tools/bench_loop.py --frames 300 > gpurun_out/loop_ab.out 2> gpurun_out/loop_ab.err || { tail -3 gpurun_out/loop_ab.err; exit 1; }
  sed -n 1,2p gpurun_out/loop_ab.out
done
timeout -k 10 200 examples/offline_fuse --n 512 --frames 300 --track > gpurun_out/offline.out 2>&1 || { tail -5 gpurun_out/offline.out; exit 1; }
tail -2 gpurun_out/offline.out
timeout -k 10 200 python tools/bench_raycast.py > gpurun_out/rc.out 2>&1 || { tail -5 gpurun_out/rc.out; exit 1; }
tail -12 gpurun_out/rc.out
