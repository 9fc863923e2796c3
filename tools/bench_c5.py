#!/usr/bin/env python3
"""Config 5 of BASELINE.json by one rank's share: a z-slab of a 2048^3 volume (2048 x 2048 x 256 voxels, 4 GiB),
four cameras of the S2 orbit rig (90 degrees apart) folded in one sweep (yakhip_integrate_batch), then GPU
marching cubes over the slab.  One GPU plays rank R of W (GPU only; no oracle: parity of these paths at this
size is covered by tests/test_integrate_gpu.py and tests/test_mc_gpu.py).

  python tools/bench_c5.py [--world 8] [--rank 3] [--reps 5]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2048)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=3)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--z-chunk", type=int, default=0, help="layers per workgroup walk (0 = the library's choice)")
    args = ap.parse_args()
    import torch

    from yak_amd import capi, synth
    from yak_amd.shard import slab_range

    ctx = capi.Context(0)
    if args.z_chunk:
        ctx.set_option("z_chunk", args.z_chunk)
    n = args.n
    dims, vs, tr = synth.volume_geometry(n)
    z0, z1 = slab_range(n, args.world, args.rank)
    z1h = min(z1 + 1, n)  # one layer of overlap for marching cubes (include/yakhip.h)
    layers = z1h - z0
    vol = torch.zeros(layers * n * n, dtype=torch.int32, device="cuda")
    own = capi.volume_desc(vol.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT, z0, z1)
    held = capi.volume_desc(vol.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT, z0, z1h)
    frames, keep = [], []
    for f in (0, 75, 150, 225):
        depth, R, t = synth.scene_s2(f)
        d = torch.from_numpy(depth.view(np.int16)).cuda()
        dd = torch.empty_like(d)
        ctx.compute_dists(d.data_ptr(), synth.COLS * 2, dd.data_ptr(), synth.COLS * 2, synth.ROWS, synth.COLS, synth.INTR)
        keep.append(dd)
        frames.append((dd.data_ptr(), synth.COLS * 2, synth.ROWS, synth.COLS, R.reshape(-1), t, synth.INTR))

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    ctx.clear_volume(held)
    ms_batch = timed(lambda: ctx.integrate_batch(held, frames), args.reps)

    def sequential():
        for fr in frames:
            ctx.integrate(held, *fr)
    ctx.set_option("defer_mixed", 0)
    ms_seq = timed(sequential, args.reps)
    ctx.set_option("defer_mixed", 2)
    ms_seq2 = timed(sequential, args.reps)
    ctx.set_option("defer_mixed", 1)
    voxels = layers * n * n
    touched = int((vol.view(layers, n * n)[:, :] >> 16 != 0).sum().item())
    print("rank %d of %d of a %d^3 volume: slab [%d, %d) + 1 = %.2f GiB, voxel %.2f mm" % (
        args.rank, args.world, n, z0, z1, voxels * 4 / 2 ** 30, vs[0] * 1e3))
    print("  4 cameras, one sweep (integrate_batch): %.3f ms = %.3f ms per frame, %.0f Gvox/s visited; 4 single-frame calls: "
          "%.3f ms as one kernel each, %.3f ms as two; %d voxels touched (%.1f %%)" % (ms_batch, ms_batch / 4, 4 * voxels / ms_batch / 1e6, ms_seq,
                                                                                      ms_seq2, touched, 100.0 * touched / voxels))
    # what the wave bricks of each camera are (the single-frame kernel counts them; the sweep classifies the same bricks)
    ctx.set_option("brick_stats", 1)
    ctx.brick_stats()
    for k, fr in enumerate(frames):
        ctx.integrate(held, *fr)  # (not timed: the counting costs an atomic per wave)
        torch.cuda.synchronize()
        st = ctx.brick_stats()
        tot = max(sum(st.values()), 1)
        print("  camera %d alone: wave bricks %s" % (k, ", ".join("%s %.1f %%" % (c, 100.0 * v / tot) for c, v in sorted(st.items()))))
    ctx.set_option("brick_stats", 0)
    ntri = ctx.marching_cubes(held, 2, vs[0])
    tri = torch.empty((max(ntri, 1), 9), dtype=torch.float32, device="cuda")
    ms_mc = timed(lambda: ctx.marching_cubes(held, 2, vs[0], tri.data_ptr(), ntri), args.reps)
    print("  marching cubes over the slab: %.3f ms per call (count, scan, emit; host waits twice), %d triangles, %.0f Gvox/s"
          % (ms_mc, ntri, voxels / ms_mc / 1e6))
    ctx.close()


if __name__ == "__main__":
    main()
