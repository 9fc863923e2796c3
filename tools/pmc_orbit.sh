# usage (on the GPU box): bash tools/pmc_orbit.sh -- SQ counters of the integrate kernel on the orbit views (tools/bench_orbit_waves.py, ordinary build)
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_orbit
rm -rf $O && mkdir -p $O
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $O/sq -- python3 $R/tools/bench_orbit_waves.py --view 3 > $O/sq.log 2>&1 || exit 3
python3 - $O <<'PY'
import csv,glob,collections,sys
rows=collections.defaultdict(dict)
for f in glob.glob(sys.argv[1]+"/sq/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "integrate_kernel<4, true>" in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]]=float(r["Counter_Value"])
ids=sorted(rows)
print(len(ids),"integrate launches")
def show(name, sel):
    keys=sorted(rows[sel[0]])
    print(name, " ".join("%s=%.4g"%(k, sum(rows[i][k] for i in sel)/len(sel)) for k in keys))
show("all 8 views, warm", ids[8:24])
show("view 3 (last launch)", ids[-1:])
PY
