#!/bin/bash
set -u
mkdir -p gpurun_out
step() { local name=$1 lim=$2; shift 2
  timeout -k 10 "$lim" "$@" > "gpurun_out/$name.out" 2> "gpurun_out/$name.err"; local rc=$?
  echo "$name rc=$rc"; if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "stopping after $name rc=$rc"; exit $rc; fi; }
step parity 900 python -u -m pytest tests/test_integrate_gpu.py tests/test_integrate_batch_gpu.py tests/test_zz_no_alloc.py -x -q --timeout 300 --timeout-method thread
tail -15 gpurun_out/parity.out
step c5 300 python tools/bench_c5.py
tail -5 gpurun_out/c5.out
bash tools/prof_loop.sh > gpurun_out/prof_loop_stdout.txt 2>&1; echo "prof_loop rc=$?"
tail -30 gpurun_out/prof_loop_stdout.txt
