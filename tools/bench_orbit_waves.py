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
    simd = (hw >> 4) & 0xfff
    busy = {}
    for s_, t_ in zip(simd, total):
        busy[int(s_)] = busy.get(int(s_), 0) + int(t_)
    b = np.array(list(busy.values()), dtype=np.float64)
    print("  wave-cycles per SIMD id (%d ids seen, the 8 XCDs share them): mean %.0f, max %.0f" % (len(b), b.mean(), b.max()))
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
    busy_us = real[m].sum() / 100.0
    print("  mixed wave time in all: %.0f us = %.1f us per SIMD if spread over 1024" % (busy_us, busy_us / 1024))
    ctx.close()


if __name__ == "__main__":
    main()
