#!/bin/bash
# one gpurun call: no-alloc test, then launch-series experiments (each step under its own limit;
# stops at the first step that ends by signal / time limit / abort)
set -u
mkdir -p gpurun_out
step() { # name, limit, command...
  local name=$1 lim=$2; shift 2
  timeout -k 10 "$lim" "$@" > "gpurun_out/$name.out" 2> "gpurun_out/$name.err"
  local rc=$?
  echo "$name rc=$rc"
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "stopping after $name"; exit $rc; fi
}
step wip 300 python -u -m pytest tests/wip_no_alloc.py -x -v --timeout 120 --timeout-method thread
for pre in none selftest spin sleep; do
  step series_$pre 120 python -u tools/launch_series.py --pre $pre
done
step series_gap 120 python -u tools/launch_series.py --pre none --gap-ms 2 --launches 100
step series_s2 120 python -u tools/launch_series.py --pre none --scene s2
step bench20 200 python bench.py --steps 20 --warmup 5 --no-rows --no-cpu-baseline
tail -3 gpurun_out/wip.out
