# usage (on the GPU box): bash tools/trace_c5.sh [rank] -- kernel trace of bench.py --config c5 --as-rank 8:R: average duration per kernel
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
RANK=${1:-3}
O=$R/gpurun_out/trace_c5
rm -rf $O && mkdir -p $O
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $R/bench.py --config c5 --as-rank 8:$RANK --steps 10 --warmup 3 --no-cpu-baseline > $O/t.log 2>&1 || exit 3
python3 - $O <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+"/t/**/*kernel_stats.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r["TotalDurationNs"]) > 2e5:
            print("%-70s calls %5s avg %9.1f us  total %8.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
