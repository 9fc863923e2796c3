#!/usr/bin/env python3
"""Per-launch device time of integrate_kernel from the start of a fresh process (the
driver's --steps 20 run measured 0.198 ms per launch, a 200-step run 0.187 -- which launches are the slow ones?).

  python tools/launch_series.py [--launches 400] [--pre none|selftest|clear|sleep] [--scene s0]

Same set-up as bench.py at N = 1 (512^3, S0), then `launches` integrate calls with the library's event pair
around each kernel; prints one JSON line: per-launch ms, start offsets, and means over windows."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--launches", type=int, default=400)
    ap.add_argument("--pre", default="none")
    ap.add_argument("--scene", default="s0")
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--gap-ms", type=float, default=0.0, help="host sleep between launches (idle gaps)")
    args = ap.parse_args()
    import torch

    from bench import wall_scene
    from yak_amd import capi, synth

    t_start = time.perf_counter()
    ctx = capi.Context(0)
    n = args.n
    dims, vs, tr = synth.volume_geometry(n)
    if args.scene == "s2":
        depth, R, t = synth.scene_s2(40)
    else:
        depth, R, t = wall_scene(dims, vs[0])
    rows, cols = depth.shape
    intr = synth.INTR
    d_depth = torch.from_numpy(depth.view(np.int16)).cuda()
    d_dists = torch.empty_like(d_depth)
    ctx.compute_dists(d_depth.data_ptr(), cols * 2, d_dists.data_ptr(), cols * 2, rows, cols, intr)
    vol = torch.empty(n ** 3, dtype=torch.int32, device="cuda")
    desc = capi.volume_desc(vol.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT)
    ctx.clear_volume(desc)
    pre = {}
    for what in args.pre.split(","):
        t0 = time.perf_counter()
        if what == "selftest":
            pre["selftest"] = ctx.selftest_division()
        elif what == "clear":
            for _ in range(20):
                ctx.clear_volume(desc)
        elif what == "sleep":
            torch.cuda.synchronize()
            time.sleep(2.0)
        elif what == "spin":  # 100 ms of unrelated device work
            x = torch.empty(1 << 28, dtype=torch.int32, device="cuda")
            for _ in range(200):
                x.add_(1)
            del x
        torch.cuda.synchronize()
        pre[what + "_s"] = round(time.perf_counter() - t0, 4)
    ctx.set_option("kernel_timing", 1)
    Rf = R.reshape(-1)
    t0 = time.perf_counter()
    for _ in range(args.launches):
        ctx.integrate(desc, d_dists.data_ptr(), cols * 2, rows, cols, Rf, t, intr)
        if args.gap_ms > 0:
            torch.cuda.synchronize()
            time.sleep(args.gap_ms * 1e-3)
    ms, st = ctx.kernel_times(args.launches)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ctx.set_option("kernel_timing", 0)
    ms = np.array(ms)

    def win(a, b):
        return round(float(ms[a:b].mean()), 5) if b <= len(ms) else None

    out = {"pre": args.pre, "scene": args.scene, "launches": len(ms), "setup_s": round(t0 - t_start, 3),
           "wall_ms_per_launch": round(wall / len(ms) * 1e3, 5),
           "mean_0_5": win(0, 5), "mean_5_25": win(5, 25), "mean_25_45": win(25, 45), "mean_45_100": win(45, 100),
           "mean_100_200": win(100, 200), "mean_200_400": win(200, 400), "min": round(float(ms.min()), 5),
           "max": round(float(ms.max()), 5),
           "first_60_ms": [round(float(v), 4) for v in ms[:60]],
           "first_60_start_ms": [round(float(v), 3) for v in st[:60]], "pre_info": pre}
    print(json.dumps(out), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
