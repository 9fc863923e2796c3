#!/bin/bash
# yakhip_integrate_prepare: the loop with and without it, three times each (stage times on the second line)
set -u
mkdir -p gpurun_out
for i in 1 2 3; do
  for a in "" "--no-prepare"; do
    timeout -k 10 200 python tools/bench_loop.py --frames 300 $a > gpurun_out/loop_ab.out 2> gpurun_out/loop_ab.err || { tail -3 gpurun_out/loop_ab.err; exit 1; }
    echo "[$a]"; sed -n 1,2p gpurun_out/loop_ab.out
  done
done
