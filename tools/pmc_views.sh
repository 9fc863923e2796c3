# usage (on the GPU box): bash tools/pmc_views.sh "<setting>" [counters...] -- SQ counters per integrate kernel on the eight
# orbit views (tools/bench_views.py --orbit), averaged per launch
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
S="$1"; shift
C="${*:-SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES}"
O=$R/gpurun_out/pmc_views
rm -rf $O && mkdir -p $O
timeout -s KILL 150 rocprofv3 --pmc $C --output-format csv -d $O/sq -- python3 $R/tools/bench_views.py --orbit "$S" > $O/sq.log 2>&1 || exit 3
python3 - $O "$S" <<'PY'
import csv,glob,collections,sys
rows=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+"/sq/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if "integrate" in n:
            rows[n.split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("setting:", sys.argv[2] or "(default)")
for n in sorted(rows):
    d=rows[n]
    print(n, len(next(iter(d.values()))), "launches:", " ".join("%s=%.4g"%(k, sum(v[-40:])/len(v[-40:])) for k,v in sorted(d.items())))
PY
tail -1 $O/sq.log
