#!/usr/bin/env python3
"""Times marching cubes (count+scan, emit) at 512^3 on the fused orbit scene for several tile depths."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    from yak_amd import capi, synth

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
    ntri = ctx.marching_cubes(desc, 2, vs[0])
    tri = torch.empty((ntri, 9), dtype=torch.float32, device="cuda")
    for zc in [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "32,16,8,4".split(","))]:
        ctx.set_option("mc_z_chunk", zc)
        ctx.marching_cubes(desc, 2, vs[0], tri.data_ptr(), ntri)
        ctx.set_option("kernel_timing", 1)
        for _ in range(5):
            ctx.marching_cubes(desc, 2, vs[0])
        count_ms, k = ctx.kernel_time()
        for _ in range(5):
            ctx.marching_cubes(desc, 2, vs[0], tri.data_ptr(), ntri)
        both_ms, k2 = ctx.kernel_time()
        ctx.set_option("kernel_timing", 0)
        print("z_chunk %3d: count+scan %.4f ms, emit %.4f ms, %d triangles" % (zc, count_ms / 5, (both_ms - count_ms) / 5, ntri),
              flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
