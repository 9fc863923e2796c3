#!/bin/bash
set -u
mkdir -p gpurun_out
step() { local name=$1 lim=$2; shift 2
  timeout -k 10 "$lim" "$@" > "gpurun_out/$name.out" 2> "gpurun_out/$name.err"; local rc=$?
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "stopping after $name rc=$rc"; tail -5 gpurun_out/$name.out gpurun_out/$name.err; exit $rc; fi; }
step trk 900 python -u -m pytest tests/test_tracking_gpu.py tests/test_host_cpp.py tests/test_sharded_kinfu_gpu.py -x -q --timeout 300 --timeout-method thread
tail -4 gpurun_out/trk.out
for i in 1 2; do
step loop_poll$i 300 python tools/bench_loop.py --frames 300
head -2 gpurun_out/loop_poll$i.out
step loop_nopoll$i 300 python tools/bench_loop.py --frames 300 --no-poll
head -2 gpurun_out/loop_nopoll$i.out
done
