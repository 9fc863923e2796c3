#!/usr/bin/env python3
"""Time of yakhip_raycast_slab per rank (one GPU plays each rank in turn) against the whole-volume ray cast, 512^3
fused orbit scene, 640x480.  python tools/bench_slab_raycast.py [world]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    from yak_amd import capi, synth
    from yak_amd.shard import held_range, slab_range

    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    ctx = capi.Context(0)
    n = 512
    dims, vs, tr = synth.volume_geometry(n)
    vol = torch.zeros(n ** 3, dtype=torch.int32, device="cuda")
    desc = capi.volume_desc(vol.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT)
    rows, cols, intr = synth.ROWS, synth.COLS, synth.INTR
    frames, keep = [], []
    for f in range(0, 300, 38):
        depth, R, t = synth.scene_s2(f)
        d = torch.from_numpy(depth.view(np.int16)).cuda()
        dd = torch.empty_like(d)
        ctx.compute_dists(d.data_ptr(), cols * 2, dd.data_ptr(), cols * 2, rows, cols, intr)
        keep.append(dd)
        frames.append((dd.data_ptr(), cols * 2, rows, cols, R.reshape(-1), t, intr))
    ctx.integrate_batch(desc, frames)
    halo = capi.raycast_halo(desc)
    pts = torch.empty((rows, cols, 4), dtype=torch.float32, device="cuda")
    nrm = torch.empty_like(pts)
    keys = torch.empty((rows, cols), dtype=torch.int32, device="cuda")
    plane = n * n

    def timed(fn, reps=20):
        fn()
        ctx.set_option("kernel_timing", 1)
        for _ in range(reps):
            fn()
        ms, _ = ctx.kernel_time()
        ctx.set_option("kernel_timing", 0)
        return ms / reps

    for view in (0, 100, 222):
        _, R_v2c, t_v2c = synth.scene_s2(view)
        Rc = np.ascontiguousarray(R_v2c.T)
        tc = (-(R_v2c.astype(np.float64).T @ t_v2c.astype(np.float64))).astype(np.float32)
        whole = timed(lambda: ctx.raycast_points(desc, Rc.reshape(-1), tc, R_v2c.reshape(-1), intr, pts.data_ptr(), cols * 16,
                                                 nrm.data_ptr(), cols * 16, rows, cols))
        per_rank = []
        for r in range(world):
            own = slab_range(n, world, r)
            h0, h1 = held_range(n, own[0], own[1], halo)
            held = capi.volume_desc(vol.data_ptr() + 4 * plane * h0, dims, vs, tr, synth.MAX_WEIGHT, h0, h1)
            per_rank.append(timed(lambda: ctx.raycast_slab(held, own[0], own[1], Rc.reshape(-1), tc, R_v2c.reshape(-1), intr,
                                                           keys.data_ptr(), cols * 4, nrm.data_ptr(), cols * 16, rows, cols,
                                                           points_ptr=pts.data_ptr(), points_pitch=cols * 16)))
        print("view %3d: whole volume %.4f ms; %d slabs (halo %d): slowest rank %.4f ms, mean %.4f ms" % (
            view, whole, world, halo, max(per_rank), sum(per_rank) / world), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
