#!/bin/bash
# usage: tools/prof_s2.sh  (on the GPU box): SQ counters of the integrate kernel on an S2 orbit view
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_s2
rm -rf $O && mkdir -p $O
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/sq -- python3 $R/bench.py --scene s2 --no-cpu-baseline --no-rows --steps 5 --warmup 2 > $O/sq.log 2>&1 || exit 3
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $O/sq2 -- python3 $R/bench.py --scene s2 --no-cpu-baseline --no-rows --steps 5 --warmup 2 > $O/sq2.log 2>&1 || exit 4
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --scene s2 --no-cpu-baseline --no-rows --steps 20 --warmup 2 > $O/trace.log 2>&1 || exit 5
python3 - <<PY
import csv,glob,collections
d=collections.defaultdict(list)
for f in glob.glob("$O/sq*/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "integrate_kernel" in r["Kernel_Name"]:
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(d.items()): print(k, "%.4g" % (sum(v)/len(v)), len(v))
for f in glob.glob("$O/trace/**/*kernel_stats.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "integrate" in r["Name"] or "prepass" in r["Name"]: print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
tail -2 $O/trace.log
