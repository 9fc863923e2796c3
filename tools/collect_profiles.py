#!/usr/bin/env python3
"""usage: tools/collect_profiles.py <tag>   (here, after `gpurun -- bash tools/profile_round.sh <tag>`)
Copies the judged summaries of gpurun_out/prof_<tag>/ into profiles/<tag>_* and rewrites profiles/traffic.json
(HBM bytes per launch of the kernels bench.py prices) from the PMC passes."""
import collections
import csv
import glob
import json
import os
import shutil
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")


def one(pattern):
    return max(glob.glob(os.path.join(src, pattern), recursive=True), key=os.path.getmtime)  # (gpurun merges: older runs' files stay)


shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats.csv"))
shutil.copy(one("pmc_fetch/**/*counter_collection.csv"), os.path.join(dst, tag + "_pmc_fetch.csv"))
shutil.copy(one("pmc_write/**/*counter_collection.csv"), os.path.join(dst, tag + "_pmc_write.csv"))
shutil.copy(one("pmc_sq/**/*counter_collection.csv"), os.path.join(dst, tag + "_pmc_sq.csv"))
shutil.copy(os.path.join(src, "summary.txt"), os.path.join(dst, tag + "_summary.txt"))
shutil.copy(os.path.join(src, "trace_bench.json"), os.path.join(dst, tag + "_bench_under_rocprof.json"))
if os.path.exists(os.path.join(src, "kernel_stats_short.csv")):  # the driver's arguments: --steps 20 --warmup 5
    shutil.copy(os.path.join(src, "kernel_stats_short.csv"), os.path.join(dst, tag + "_kernel_stats_steps20.csv"))
    shutil.copy(os.path.join(src, "trace_short_bench.json"), os.path.join(dst, tag + "_bench_steps20_under_rocprof.json"))


def mean_kib(path, counter, kernel):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and kernel in r["Kernel_Name"]]
    return statistics.mean(v), len(v)


fetch, write = os.path.join(dst, tag + "_pmc_fetch.csv"), os.path.join(dst, tag + "_pmc_write.csv")
how = ("profiles/%s_pmc_fetch.csv + profiles/%s_pmc_write.csv (separate rocprofv3 --pmc passes of `bench.py --steps 5 --warmup 2`); "
       "FETCH_SIZE (KiB) doubled: on gfx950 the counter reports half the bytes of 64-byte requests (MI355X_MICROARCH.md, HBM section; "
       "calibrated with tools/fetch_calib.hip: a 512 MiB buffer read once reports 262147 KiB)" % (tag, tag))
out = collections.OrderedDict()
try:  # entries other scripts added (tools/collect_c5_traffic.py) stay
    out.update(json.load(open(os.path.join(dst, "traffic.json")), object_pairs_hook=collections.OrderedDict))
except (OSError, ValueError):
    pass
f, n = mean_kib(fetch, "FETCH_SIZE", "integrate_kernel<4, true>")
w, _ = mean_kib(write, "WRITE_SIZE", "integrate_kernel<4, true>")
out["integrate_s0_n1"] = {"kernel": "yakhip::integrate_kernel<4, true>", "launches": n, "FETCH_SIZE_KiB": round(f), "WRITE_SIZE_KiB": round(w),
                          "fetch_bytes_corrected": int(round(f) * 2048), "write_bytes": int(round(w) * 1024),
                          "hbm_bytes_per_launch": int(round(f) * 2048 + round(w) * 1024), "source": how}
for key, kern, what in (("raycast_s2_n1", "raycast_kernel<true, false, true, 1>", "with the surface map"),
                        ("raycast_s2_n1_without_surface_map", "raycast_kernel<true, false, false, 1>", "without the surface map")):
    f, n = mean_kib(fetch, "FETCH_SIZE", kern)
    w, _ = mean_kib(write, "WRITE_SIZE", kern)
    out[key] = {"kernel": "yakhip::" + kern, "what": "mean over the five orbit views bench.py times, " + what, "launches": n,
                "FETCH_SIZE_KiB": round(f), "WRITE_SIZE_KiB": round(w), "fetch_bytes_corrected": int(round(f) * 2048),
                "write_bytes": int(round(w) * 1024), "hbm_bytes_per_launch": int(round(f) * 2048 + round(w) * 1024), "source": how}
f, n = mean_kib(fetch, "FETCH_SIZE", "mc_kernel<false>")
out["marching_cubes_count_s2_n1"] = {"kernel": "yakhip::mc_kernel<false>", "launches": n, "FETCH_SIZE_KiB": round(f),
                                     "fetch_bytes_corrected": int(round(f) * 2048), "volume_bytes": 512 ** 3 * 4, "source": how}
json.dump(out, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print(json.dumps({k: v.get("hbm_bytes_per_launch", v.get("fetch_bytes_corrected")) for k, v in out.items()}, indent=1))
