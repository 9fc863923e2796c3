#!/bin/bash
# the C++ example over the 300-frame orbit at 512^3, with supplied poses and tracked, twice each
set -u
mkdir -p gpurun_out
for i in 1 2; do
  for a in "" "--track"; do
    timeout -k 10 200 examples/offline_fuse --n 512 --frames 300 $a > gpurun_out/offline.out 2>&1 || { tail -5 gpurun_out/offline.out; exit 1; }
    grep "frames/s" gpurun_out/offline.out
  done
done
