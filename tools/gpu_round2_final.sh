#!/bin/bash
# round-2 closing run: whole GPU suite, smoke, the driver's bench command, the config-5 rehearsal
set -u
mkdir -p gpurun_out
step() { local name=$1 lim=$2; shift 2
  timeout -k 10 "$lim" "$@" > "gpurun_out/$name.out" 2> "gpurun_out/$name.err"; local rc=$?
  if [ $rc -ne 0 ]; then echo "stopping after $name rc=$rc"; tail -15 gpurun_out/$name.out; tail -5 gpurun_out/$name.err; exit $rc; fi; }
step suite 1100 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread
tail -3 gpurun_out/suite.out
step smoke 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
tail -1 gpurun_out/smoke.out
step bench_driver 600 python bench.py --steps 20 --warmup 5
step bench_default 600 python bench.py
step bench_c5 600 python bench.py --config c5 --as-rank 8:3
for f in bench_driver bench_default bench_c5; do tail -1 gpurun_out/$f.out | cut -c1-600; done
