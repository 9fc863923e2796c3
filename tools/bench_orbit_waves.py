#!/usr/bin/env python3
"""Where the integrate kernel's time goes on a surface-rich frame: per-wave shader-clock readings of one launch
(a library built with `make EXTRA=-DYAK_INTEGRATE_TIMING`; GPU only).  Usage: python tools/bench_orbit_waves.py [--view 3]"""
import argparse
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--view", type=int, default=3)
    ap.add_argument("--n", type=int, default=512)
    args = ap.parse_args()
    import torch

    from yak_amd import capi, synth

    try:
        rows_fn = capi.load().yakhip_debug_integrate_rows
        rows_fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    except AttributeError:
        rows_fn = None  # (an ordinary build: the launches only, e.g. under rocprofv3 --pmc)
    ctx = capi.Context(0)
    n = args.n
    rows, cols, intr = synth.ROWS, synth.COLS, synth.INTR
    dims, vs, tr = synth.volume_geometry(n)
    vol = torch.zeros(n ** 3, dtype=torch.int32, device="cuda")
    desc = capi.volume_desc(vol.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT)
    views, keep = [], []
    for f in range(0, 300, 38):
        depth, R, t = synth.scene_s2(f, noise_mm=1.0)
        d = torch.from_numpy(depth.view(np.int16)).cuda()
        dd = torch.empty_like(d)
        ctx.compute_dists(d.data_ptr(), cols * 2, dd.data_ptr(), cols * 2, rows, cols, intr)
        keep.append((d, dd))
        views.append((dd.data_ptr(), R.reshape(-1), t))
    for _ in range(3):
        for ptr, R, t in views:
            ctx.integrate(desc, ptr, cols * 2, rows, cols, R, t, intr)
    torch.cuda.synchronize()
    out = np.zeros((1 << 17, 6), dtype=np.uint64)
    if rows_fn:
        rows_fn(out.ctypes.data, 1 << 17)                               # (clears the rows)
    ptr, R, t = views[args.view]
    ctx.set_option("kernel_timing", 1)
    ctx.integrate(desc, ptr, cols * 2, rows, cols, R, t, intr)
    ms, k = ctx.kernel_time()
    ctx.set_option("kernel_timing", 0)
    if not rows_fn:
        print("view %d: kernel %.4f ms (no per-wave rows: the library was not built with -DYAK_INTEGRATE_TIMING)" % (args.view, ms / k))
        ctx.close()
        return
    ph = np.zeros((1 << 17, 8), dtype=np.uint64)
    try:
        ph_fn = capi.load().yakhip_debug_integrate_phases
        ph_fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
        ph_fn(ph.ctypes.data, 1 << 17)
    except AttributeError:
        ph_fn = None
    got = rows_fn(out.ctypes.data, 1 << 17)
    slots = np.nonzero(out[:got, 0] > 0)[0]
    r = out[:got][out[:got, 0] > 0].astype(np.int64)
    cls = r[:, 0] - 1                                                     # 0 mixed, 1 dead, 2 front
    total, to_cls, replay, layers = r[:, 1], r[:, 2], r[:, 3], r[:, 4] & 0xffff
    real = r[:, 4] >> 16                                                  # duration in 10 ns ticks
    start = r[:, 5] & 0xffffffffff
    hw = r[:, 5] >> 40
    start = (start - start.min()) & 0xffffffffff
    span = int((start + real).max())
    print("view %d: kernel %.4f ms; %d waves reported, launch spans %.1f us" % (args.view, ms / k, len(r), span / 100.0))
    for c, name in ((1, "dead"), (2, "front"), (0, "mixed")):
        m = cls == c
        if not m.any():
            continue
        print("  %-5s %6d waves: cycles mean %7.0f max %7d, sum %5.1f M; brick test mean %5.0f; replay mean %6.0f (%.0f %% of these waves' cycles); layers mean %.1f"
              % (name, m.sum(), total[m].mean(), total[m].max(), total[m].sum() / 1e6, to_cls[m].mean(), replay[m].mean(),
                 100.0 * replay[m].sum() / max(total[m].sum(), 1), layers[m].mean()))
    # when does the work end?  waves still running in the last tenth of the launch, by class
    late = start + real > 0.9 * span
    print("  waves that end in the last tenth of the launch: %d mixed, %d front, %d dead; started there: %d"
          % ((late & (cls == 0)).sum(), (late & (cls == 2)).sum(), (late & (cls == 1)).sum(), (start > 0.9 * span).sum()))
    # per SIMD (HW_ID: wave 3:0, simd 5:4, cu 11:8, sh 12, se 15:13 on gfx9; plus the XCC from the workgroup id is not in it)
    # (bits 16..19 of the recorded word: the XCC id, so a SIMD id is unique on the chip)
    simd = (hw >> 4) & 0xffff
    work = replay_steps_instr = None
    busy, load, first, last, nmix = {}, {}, {}, {}, {}
    est = np.where(cls == 0, layers * 260 + (replay // 1), 200)  # rough instruction estimate: walk + replay cycles as they are
    for s_, t_, e_, st_, du_, c_ in zip(simd, total, est, start, real, cls):
        s_ = int(s_)
        busy[s_] = busy.get(s_, 0) + int(t_)
        first[s_] = min(first.get(s_, 1 << 60), int(st_))
        last[s_] = max(last.get(s_, 0), int(st_ + du_))
        nmix[s_] = nmix.get(s_, 0) + (1 if c_ == 0 else 0)
    b = np.array(list(busy.values()), dtype=np.float64)
    print("  wave-cycles per SIMD (%d SIMDs seen): mean %.0f, max %.0f" % (len(b), b.mean(), b.max()))
    ends = np.array([last[k] for k in busy], dtype=np.float64) / 100.0
    nm = np.array([nmix[k] for k in busy], dtype=np.float64)
    print("  per SIMD: last wave ends at us: min %.1f  10%% %.1f  median %.1f  90%% %.1f  max %.1f" %
          (ends.min(), np.percentile(ends, 10), np.median(ends), np.percentile(ends, 90), ends.max()))
    print("  per SIMD: mixed waves: min %d  mean %.2f  max %d;  correlation(end time, mixed waves) %.2f" %
          (nm.min(), nm.mean(), nm.max(), np.corrcoef(ends, nm)[0, 1]))
    # the same per CU (4 SIMDs)
    cu_end, cu_mix = {}, {}
    for k in busy:
        c_ = k >> 2
        cu_end[c_] = max(cu_end.get(c_, 0), last[k])
        cu_mix[c_] = cu_mix.get(c_, 0) + nmix[k]
    ce = np.array(list(cu_end.values()), dtype=np.float64) / 100.0
    cm = np.array([cu_mix[k] for k in cu_end], dtype=np.float64)
    print("  per CU (%d seen): end us min %.1f median %.1f max %.1f; mixed waves min %d mean %.1f max %d" %
          (len(ce), ce.min(), np.median(ce), ce.max(), cm.min(), cm.mean(), cm.max()))
    # start times: how long until the last wave starts
    print("  last wave starts at %.0f %% of the launch; mixed waves start at %.0f %% on average"
          % (100.0 * start.max() / span, 100.0 * start[cls == 0].mean() / span if (cls == 0).any() else 0))
    m = cls == 0
    order = np.argsort(-(start[m] + real[m]))
    print("  the mixed waves that end last (start us, duration us, cycles, replay cycles, layers):")
    for q in order[:8]:
        print("     %6.1f %6.1f %7d %7d %3d" % (start[m][q] / 100.0, real[m][q] / 100.0, total[m][q], replay[m][q], layers[m][q]))
    hist, edges = np.histogram(start[m] / 100.0, bins=8, range=(0, span / 100.0))
    print("  mixed waves by start time (eighths of the launch):", hist.tolist())
    hist, edges = np.histogram((start[m] + real[m]) / 100.0, bins=8, range=(0, span / 100.0))
    print("  mixed waves by end time:", hist.tolist())
    # per z-chunk, in dispatch order (blockIdx.y): rows are indexed ((chunk * blocks + block) * 4 + wave)
    per_chunk = 4 * ((n * n // 4 + 255) // 256)
    chunk = slots // per_chunk
    print("  chunk (dispatch order): waves mixed / front / dead, mixed wave-time us, first start us, last end us")
    for c in range(int(chunk.max()) + 1):
        k = chunk == c
        km = k & (cls == 0)
        print("    %2d: %4d %4d %5d  %8.0f  %6.1f %6.1f  layers %d" % (c, km.sum(), (k & (cls == 2)).sum(), (k & (cls == 1)).sum(), real[km].sum() / 100.0,
                                                   start[k].min() / 100.0, (start[k] + real[k]).max() / 100.0, layers[km].max() if km.any() else 0))
    if ph_fn:
        pm = ph[:got][out[:got, 0] > 0].astype(np.int64)[m]
        tot_m = total[m].astype(np.float64)
        names = ["stage 1", "settle", "wait for voxels", "fuse", "store + look-ahead"]
        print("  mixed waves, share of their cycles: brick test %.1f %%, replay %.1f %%, " % (100.0 * to_cls[m].sum() / tot_m.sum(), 100.0 * replay[m].sum() / tot_m.sum())
              + ", ".join("%s %.1f %%" % (nm, 100.0 * pm[:, i].sum() / tot_m.sum()) for i, nm in enumerate(names)))
        lay = np.maximum(layers[m], 1)
        print("  per walked layer: cycles %.0f; waits for voxels %.2f, per-voxel path %.2f of the layers" %
              ((tot_m - to_cls[m] - replay[m]).sum() / lay.sum(), pm[:, 5].sum() / lay.sum(), pm[:, 6].sum() / lay.sum()))
        slow = np.argsort(-tot_m)[:6]
        for q in slow:
            print("     slowest: cycles %7d layers %2d brick test %6d replay %6d | " % (tot_m[q], layers[m][q], to_cls[m][q], replay[m][q])
                  + " ".join("%s %6d" % (nm, pm[q, i]) for i, nm in enumerate(names)) + " | waits %d detailed %d" % (pm[q, 5], pm[q, 6]))
    busy_us = real[m].sum() / 100.0
    print("  mixed wave time in all: %.0f us = %.1f us per SIMD if spread over 1024" % (busy_us, busy_us / 1024))
    ctx.close()


if __name__ == "__main__":
    main()
