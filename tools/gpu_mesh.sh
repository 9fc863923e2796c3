#!/bin/bash
# the C++ host tests, then offline_fuse with the laps of FusionServer::extractMesh (YAK_MC_TIMING=1)
set -u
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_host_cpp.py -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/host_tests.out 2>&1 || { tail -20 gpurun_out/host_tests.out; exit 1; }
tail -1 gpurun_out/host_tests.out
YAK_MC_TIMING=1 timeout -k 10 200 examples/offline_fuse --n 512 --frames 300 > gpurun_out/offline.out 2>&1 || { tail -5 gpurun_out/offline.out; exit 1; }
grep -i "marchingCubesDevice\|extractMesh\|frames/s" gpurun_out/offline.out | cut -c1-200
