#!/bin/bash
# usage: tools/prof_mc.sh <tag>  (on the GPU box): kernel times and FETCH_SIZE of the marching-cubes passes
set -u
tag=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/mc_$tag
rm -rf $O && mkdir -p $O
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/sweep_mc.py 4 > $O/trace.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc -- python3 $R/tools/sweep_mc.py 4 > $O/pmc.log 2>&1 || exit 2
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq -- python3 $R/tools/sweep_mc.py 4 > $O/pmc_sq.log 2>&1 || exit 3
python3 - <<PY
import csv,glob,collections
d=collections.defaultdict(list)
for f in glob.glob("$O/pmc/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "mc_kernel" in r["Kernel_Name"]:
            d[r["Kernel_Name"][40:70]].append(float(r["Counter_Value"]))
for k,v in d.items(): print(k, "FETCH_SIZE KiB mean", sum(v)/len(v), "x2 / volume =", 2*sum(v)/len(v)/524288)
d=collections.defaultdict(list)
for f in glob.glob("$O/pmc_sq/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "mc_kernel" in r["Kernel_Name"]:
            d[(r["Kernel_Name"][40:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k,v in sorted(d.items()): print(k, "%.4g" % (sum(v)/len(v)))
for f in glob.glob("$O/trace/**/*kernel_stats.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "mc_" in r["Name"]: print(r["Name"][38:70], r["Calls"], r["AverageNs"], r["MinNs"])
PY
