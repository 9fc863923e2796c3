#!/bin/bash
# usage: tools/prof_rank.sh W:R [W:R ...]   (on the GPU box): SQ counters of the integrate kernel for emulated ranks
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for rank in "$@"; do
  O=$R/gpurun_out/prof_rank_${rank/:/_}
  rm -rf $O && mkdir -p $O
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/sq -- python3 $R/bench.py --as-rank $rank --no-cpu-baseline --no-rows --steps 5 --warmup 2 > $O/sq.log 2>&1 || exit 3
  timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $O/tcc -- python3 $R/bench.py --as-rank $rank --no-cpu-baseline --no-rows --steps 5 --warmup 2 > $O/tcc.log 2>&1 || echo "tcc pass failed"
  python3 - $O $rank <<'PY'
import csv,glob,collections,sys
d=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+"/*/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "integrate_kernel" in r["Kernel_Name"]:
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("rank", sys.argv[2], " ".join("%s=%.4g" % (k, sum(v)/len(v)) for k,v in sorted(d.items())))
PY
done
