#!/bin/bash
set -u
mkdir -p gpurun_out
run() { timeout -k 10 200 python tools/bench_loop.py --frames 300 "$@" > gpurun_out/loop_ab.out 2>&1 || { tail -3 gpurun_out/loop_ab.out; exit 1; }; echo "$* :"; sed -n 1,2p gpurun_out/loop_ab.out; }
for i in 1 2; do run; run --no-overlap; done
run --no-icp; run --no-icp --no-overlap
