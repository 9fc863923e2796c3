#!/bin/bash
# usage: tools/prof_loop.sh   (run on the GPU box through gpurun)
# Kernel-trace statistics of the complete KinFu loop (tools/bench_loop.py): per-kernel average durations of
# the tracking-side kernels, into gpurun_out/prof_loop/.
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_loop
rm -rf $O && mkdir -p $O
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/bench_loop.py --frames 300 > $O/bench_loop.txt 2> $O/trace.log || exit 1
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python3 - $O/kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-60s calls %5s avg %9.1f ns  min %8s max %8s" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"]))
PY
cat $O/bench_loop.txt
