#!/usr/bin/env python3
"""usage: tools/collect_c5_traffic.py   (here, after `gpurun -- bash tools/pmc_c5.sh`)
Adds the HBM bytes per launch of config 5's batched integrate (emulated ranks 0 and 3 of 8) to profiles/traffic.json
and keeps the two counter files under profiles/."""
import csv
import glob
import json
import os
import shutil
import statistics

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "pmc_c5")
dst = os.path.join(ROOT, "profiles")
KERNEL = "integrate_batch_kernel<4, true>"


def counter_file(sub):
    return max(glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)


def mean_kib(path, counter):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and KERNEL in r["Kernel_Name"]]
    return statistics.mean(v), len(v)


path = os.path.join(dst, "traffic.json")
rec = json.load(open(path))
for r in (0, 3):
    fetch = os.path.join(dst, "r02_pmc_c5_rank%d_fetch.csv" % r)
    write = os.path.join(dst, "r02_pmc_c5_rank%d_write.csv" % r)
    shutil.copy(counter_file("fetch_r%d" % r), fetch)
    shutil.copy(counter_file("write_r%d" % r), write)
    f, n = mean_kib(fetch, "FETCH_SIZE")
    w, _ = mean_kib(write, "WRITE_SIZE")
    rec["integrate_batch_c5_rank%d_of_8" % r] = {
        "kernel": "yakhip::" + KERNEL, "launches": n, "FETCH_SIZE_KiB": round(f), "WRITE_SIZE_KiB": round(w),
        "fetch_bytes_corrected": int(round(f) * 2048), "write_bytes": int(round(w) * 1024),
        "hbm_bytes_per_launch": int(round(f) * 2048 + round(w) * 1024),
        "source": "profiles/r02_pmc_c5_rank%d_fetch.csv + _write.csv (separate rocprofv3 --pmc passes of `bench.py --config c5 "
                  "--as-rank 8:%d --steps 5 --warmup 2`); FETCH_SIZE (KiB) doubled as for the other entries" % (r, r)}
    print(r, rec["integrate_batch_c5_rank%d_of_8" % r]["hbm_bytes_per_launch"])
json.dump(rec, open(path, "w"), indent=1)
