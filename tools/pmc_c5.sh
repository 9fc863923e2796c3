#!/bin/bash
# usage: tools/pmc_c5.sh   (on the GPU box through gpurun)
# HBM traffic of config 5's sweep (bench.py --config c5 --as-rank 8:r), r = 0 and 3: one FETCH_SIZE and
# one WRITE_SIZE pass each, into gpurun_out/pmc_c5/; tools/collect_c5_traffic.py puts them into profiles/traffic.json
# (round 3: per launch kind -- the sweep kernel, or the four single-frame launches of two kernels the library runs instead
# where the slab is mostly surface).
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_c5
rm -rf $O && mkdir -p $O
for r in 0 3; do
  timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_r$r -- python3 $R/bench.py --config c5 --as-rank 8:$r --steps 5 --warmup 3 --no-cpu-baseline > $O/fetch_r$r.log 2>&1 || exit 1
  timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_r$r -- python3 $R/bench.py --config c5 --as-rank 8:$r --steps 5 --warmup 3 --no-cpu-baseline > $O/write_r$r.log 2>&1 || exit 2
  tail -1 $O/write_r$r.log | cut -c1-200
done
