#!/bin/bash
# usage: tools/prof_rank.sh W:R  (on the GPU box): SQ counters of the integrate kernel for an emulated rank
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_rank_${1/:/_}
rm -rf $O && mkdir -p $O
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/sq -- python3 $R/bench.py --as-rank $1 --no-cpu-baseline --no-rows --steps 5 --warmup 2 > $O/sq.log 2>&1 || exit 3
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum --output-format csv -d $O/tc -- python3 $R/bench.py --as-rank $1 --no-cpu-baseline --no-rows --steps 5 --warmup 2 > $O/tc.log 2>&1 || exit 4
python3 - <<PY
import csv,glob,collections
d=collections.defaultdict(list)
for f in glob.glob("$O/*/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "integrate_kernel" in r["Kernel_Name"]:
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$1", {k: "%.4g" % (sum(v)/len(v)) for k,v in sorted(d.items())})
PY
