#!/bin/bash
# usage: bash tools/gpu_loop_ab.sh [bench_loop.py arguments]   -- the fusion loop three times, first two lines of each
set -u
mkdir -p gpurun_out
for i in 1 2 3; do
  timeout -k 10 200 python tools/bench_loop.py --frames 300 "$@" > gpurun_out/loop_ab.out 2>&1 || { tail -3 gpurun_out/loop_ab.out; exit 1; }
  sed -n 1,2p gpurun_out/loop_ab.out
done
