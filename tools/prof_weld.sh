# usage (on the GPU box): bash tools/prof_weld.sh -- kernel trace of tools/bench_weld.py: average duration per kernel
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_weld
rm -rf $O && mkdir -p $O
timeout -k 10 200 python3 $R/tools/bench_weld.py || exit 2
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $R/tools/bench_weld.py > $O/t.log 2>&1 || exit 3
python3 - $O <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+"/t/**/*kernel_stats.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Name"]
        if any(k in n for k in ("weld","mc_","radix","onesweep","select","partition","scan","sort","lookback","histogram")):
            print("%-90s calls %5s avg %9.1f us total %9.1f us" % (n[:90], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e3))
PY
