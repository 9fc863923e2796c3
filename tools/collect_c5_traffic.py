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
SINGLES = ("classify_kernel<4>", "integrate_list_kernel<4, true>", "checkpoint_tiles_kernel<4>")
TAG = "r03"


def counter_file(sub):
    return max(glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)


def mean_kib(path, counter, kernel=KERNEL):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and kernel in r["Kernel_Name"]]
    return (statistics.mean(v), len(v)) if v else (0.0, 0)


path = os.path.join(dst, "traffic.json")
rec = json.load(open(path))
how = ("profiles/%s_pmc_c5_rank%%d_fetch.csv + _write.csv (separate rocprofv3 --pmc passes of `bench.py --config c5 --as-rank 8:%%d "
       "--steps 5 --warmup 3`); FETCH_SIZE (KiB) doubled as for the other entries" % TAG)
for r in (0, 3):
    fetch = os.path.join(dst, "%s_pmc_c5_rank%d_fetch.csv" % (TAG, r))
    write = os.path.join(dst, "%s_pmc_c5_rank%d_write.csv" % (TAG, r))
    shutil.copy(counter_file("fetch_r%d" % r), fetch)
    shutil.copy(counter_file("write_r%d" % r), write)
    f, n = mean_kib(fetch, "FETCH_SIZE")
    w, _ = mean_kib(write, "WRITE_SIZE")
    if n:  # the sweep kernel ran (at least for the first sweeps, before the library had findings)
        rec["integrate_batch_c5_rank%d_of_8" % r] = {
            "kernel": "yakhip::" + KERNEL, "launches": n, "FETCH_SIZE_KiB": round(f), "WRITE_SIZE_KiB": round(w),
            "fetch_bytes_corrected": int(round(f) * 2048), "write_bytes": int(round(w) * 1024),
            "hbm_bytes_per_launch": int(round(f) * 2048 + round(w) * 1024), "source": how % (r, r)}
        print(r, "sweep kernel", n, "launches", rec["integrate_batch_c5_rank%d_of_8" % r]["hbm_bytes_per_launch"])
    # four single-frame launches of two (deep slabs: three) kernels each: bytes per sweep = 4 x the kernels' means
    parts, total, launches = {}, 0, 0
    for k in SINGLES:
        fk, nk = mean_kib(fetch, "FETCH_SIZE", k)
        wk, _ = mean_kib(write, "WRITE_SIZE", k)
        if nk:
            parts[k] = {"launches": nk, "fetch_bytes_corrected": int(round(fk) * 2048), "write_bytes": int(round(wk) * 1024)}
            total += int(round(fk) * 2048 + round(wk) * 1024)
            launches = max(launches, nk)
    if launches:
        rec["integrate_singles_c5_rank%d_of_8" % r] = {
            "kernel": "4 x (" + " + ".join("yakhip::" + k for k in parts) + ")", "launches_of_each": launches, "per_kernel": parts,
            "hbm_bytes_per_launch": 4 * total, "what": "per sweep of four frames: four times the kernels' mean bytes per launch",
            "source": how % (r, r)}
        print(r, "singles", launches, "launches of each", 4 * total)
json.dump(rec, open(path, "w"), indent=1)
