# usage (on the GPU box): bash tools/trace_views.sh "<setting>" [bench_views flags] -- kernel trace of tools/bench_views.py: average
# duration per integrate kernel and the gap between the two kernels of a launch
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
S="$1"; shift
O=$R/gpurun_out/trace_views
rm -rf $O && mkdir -p $O
timeout -s KILL 150 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 $R/tools/bench_views.py "$@" "$S" > $O/t.log 2>&1 || exit 3
python3 - $O "$S" <<'PY'
import csv,glob,sys,collections
rows=[]
for f in glob.glob(sys.argv[1]+"/t/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]))
rows.sort()
d=collections.defaultdict(list)
gaps=[]
for i,(s,e,n) in enumerate(rows):
    if "integrate" in n or "prepass" in n or "classify" in n or "checkpoint" in n:
        d[n.split("(")[0][-50:]].append(e-s)
    if "integrate_mixed_kernel" in n and i>0 and "integrate_kernel" in rows[i-1][2]:
        gaps.append(s-rows[i-1][1])
print("setting:", sys.argv[2] or "(default)")
for n,v in sorted(d.items()):
    v=v[-40:]
    print("  %-50s %4d launches, last 40: mean %.1f us  min %.1f  max %.1f" % (n,len(d[n]),sum(v)/len(v)/1e3,min(v)/1e3,max(v)/1e3))
if gaps:
    g=gaps[-40:]
    print("  gap first -> second kernel: mean %.1f us (min %.1f, max %.1f)" % (sum(g)/len(g)/1e3,min(g)/1e3,max(g)/1e3))
PY
tail -1 $O/t.log
