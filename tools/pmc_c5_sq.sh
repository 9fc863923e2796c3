#!/bin/bash
# usage: tools/pmc_c5_sq.sh [rank]   (on the GPU box through gpurun): instruction counters of config 5's batched sweep
set -u
r=${1:-3}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_c5_sq
rm -rf $O && mkdir -p $O
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/sq -- python3 $R/bench.py --config c5 --as-rank 8:$r --steps 5 --warmup 2 --no-cpu-baseline > $O/sq.log 2>&1 || exit 1
python3 - $(find $O/sq -name "*counter_collection.csv" | head -1) <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "integrate_batch_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%-22s mean %.4g over %d launches" % (k, sum(v) / len(v), len(v)))
PY
