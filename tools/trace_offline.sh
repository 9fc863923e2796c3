#!/bin/bash
# usage: tools/trace_offline.sh [offline_fuse arguments]   (on the GPU box through gpurun)
# Kernel trace of the C++ example: where the main chain's time goes between its kernels.
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/trace_offline
rm -rf $O && mkdir -p $O
timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $O/t -- $R/examples/offline_fuse --n 512 --frames 300 "$@" > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
grep "frames/s" $O/run.log
python3 - $(find $O/t -name "*kernel_trace.csv" | head -1) <<'PY'
import csv, sys, statistics
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
main = [r for r in rows if any(k in r["Kernel_Name"] for k in ("integrate_kernel", "raycast_kernel", "icp_"))]
def short(n):
    for k in ("integrate_kernel", "raycast_kernel", "icp_accumulate", "icp_solve"):
        if k in n:
            return k
    return n[:30]
gaps = {}
for a, b in zip(main, main[1:]):
    key = short(a["Kernel_Name"]) + " -> " + short(b["Kernel_Name"])
    gaps.setdefault(key, []).append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
for k, v in gaps.items():
    v = v[5:] if len(v) > 20 else v
    print("%-44s n %4d  median gap %7.2f us  mean %7.2f" % (k, len(v), statistics.median(v), statistics.mean(v)))
dur = {}
for r in rows:
    dur.setdefault(short(r["Kernel_Name"]), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in dur.items():
    print("%-32s n %5d  mean %8.2f us" % (k, len(v), statistics.mean(v)))
ints = [r for r in rows if "integrate_kernel" in r["Kernel_Name"]]
per = [(int(b["Start_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3 for a, b in zip(ints, ints[1:])][5:]
print("integrate start to integrate start: median %.1f us" % statistics.median(per))
PY
