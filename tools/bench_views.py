#!/usr/bin/env python3
"""A/B timing of the single-frame integrate kernel at 512^3: the wall scene S0 (the roofline case), S1 and the eight orbit
views of bench.py's `orbit_views` row, for one or more settings of the context's tuning options (GPU only; no oracle).

  python tools/bench_views.py "" "tile=3" "tile=3,chunk_min=4"
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    from yak_amd import capi, synth

    ctx = capi.Context(0)
    n = 512
    rows, cols, intr = synth.ROWS, synth.COLS, synth.INTR
    dims, vs, tr = synth.volume_geometry(n)
    vol = torch.zeros(n ** 3, dtype=torch.int32, device="cuda")
    desc = capi.volume_desc(vol.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT)
    keep = []

    def frame(depth, R, t):
        d = torch.from_numpy(depth.view(np.int16)).cuda()
        dd = torch.empty_like(d)
        ctx.compute_dists(d.data_ptr(), cols * 2, dd.data_ptr(), cols * 2, rows, cols, intr)
        keep.append((d, dd))
        return (dd.data_ptr(), cols * 2, rows, cols, R.reshape(-1), t, intr)

    s0, s1 = frame(*synth.scene_s0()), frame(*synth.scene_s1())
    views = [frame(*synth.scene_s2(f, noise_mm=1.0)) for f in range(0, 300, 38)]

    sync = "--no-sync" not in sys.argv

    def timed(frames, reps):
        for a in frames:
            ctx.integrate(desc, *a)
            if sync:
                torch.cuda.synchronize()
        ctx.set_option("kernel_timing", 1)
        for _ in range(reps):
            for a in frames:
                ctx.integrate(desc, *a)
                if sync:
                    torch.cuda.synchronize()  # (one wait per frame, as a tracking loop has: the launcher's hint stays fresh)
        ms, _ = ctx.kernel_times(reps * len(frames))
        ctx.set_option("kernel_timing", 0)
        return [sum(ms[i::len(frames)]) / reps for i in range(len(frames))]

    orbit_only = "--orbit" in sys.argv
    settings = [a for a in sys.argv[1:] if a not in ("--orbit", "--no-sync")] or [""]
    defaults = {"tile": 0, "chunk_order": -1, "brick_test": 1, "group_test": 1, "tile_side": 1, "defer_mixed": 1, "refine": 1}
    # warm the board (bench.py: launches right after idle run slow)
    for _ in range(60):
        ctx.integrate(desc, *s0)
    torch.cuda.synchronize()
    for s in settings:
        opts = [kv.split("=") for kv in s.split(",") if kv]
        for k, v in opts:
            ctx.set_option(k, int(v))
        ctx.clear_volume(desc)
        t0 = t1 = 0.0
        if not orbit_only:
            t0 = timed([s0], 20)[0]
            t1 = timed([s1], 10)[0]
            ctx.clear_volume(desc)
        per = timed(views, 5)
        ctx.set_option("brick_stats", 1)
        for a in views:
            ctx.integrate(desc, *a)
        bricks = ctx.brick_stats()
        ctx.set_option("brick_stats", 0)
        tot = max(sum(bricks.values()), 1)
        print("%-28s S0 %.4f  S1 %.4f  orbit mean %.4f  [%s]  bricks %s" %
              (s or "(default)", t0, t1, sum(per) / len(per), " ".join("%.3f" % v for v in per),
               {k: round(v / tot, 3) for k, v in bricks.items()}), flush=True)
        for k, v in opts:  # back to the defaults
            ctx.set_option(k, defaults.get(k, 0))
    ctx.close()


if __name__ == "__main__":
    main()
