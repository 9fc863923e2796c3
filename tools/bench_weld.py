#!/usr/bin/env python3
"""Times yakhip_marching_cubes + yakhip_weld_vertices on a 512^3 volume fused from eight views of the S2 orbit (GPU only).
Run it under `rocprofv3 --kernel-trace --stats` for the kernels of a call."""
import os
import sys
import time

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
    for f in range(0, 300, 38):
        depth, R, t = synth.scene_s2(f, noise_mm=1.0)
        d = torch.from_numpy(depth.view(np.int16)).cuda()
        dd = torch.empty_like(d)
        ctx.compute_dists(d.data_ptr(), cols * 2, dd.data_ptr(), cols * 2, rows, cols, intr)
        ctx.integrate(desc, dd.data_ptr(), cols * 2, rows, cols, R.reshape(-1), t, intr)
    torch.cuda.synchronize()
    ntri = ctx.marching_cubes(desc, 4, 1.0)
    tri = torch.empty((ntri, 9), dtype=torch.float32, device="cuda")
    ctx.marching_cubes(desc, 4, 1.0, tri.data_ptr(), ntri)
    wv = torch.empty((ntri * 3, 3), dtype=torch.float32, device="cuda")
    wi = torch.empty(ntri * 3, dtype=torch.int32, device="cuda")
    nvert = ctx.weld_vertices(tri.data_ptr(), ntri, wv.data_ptr(), wi.data_ptr())
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.weld_vertices(tri.data_ptr(), ntri, wv.data_ptr(), wi.data_ptr())
    tw = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.marching_cubes(desc, 4, 1.0, tri.data_ptr(), ntri)
    tm = (time.perf_counter() - t0) / reps
    print("%d triangles, %d vertices; weld %.3f ms per blocking call, marching cubes %.3f ms per blocking call" %
          (ntri, nvert, tw * 1e3, tm * 1e3), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
