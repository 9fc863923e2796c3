#!/usr/bin/env python3
"""Times the single-frame integrate kernel at 512^3 on scenes S0 / S1 / S2 for every wave footprint
("tile" option) and a few z-chunk sizes (GPU only; no oracle)."""
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
    scenes = [("S0", synth.scene_s0()), ("S1", synth.scene_s1()), ("S2/0", synth.scene_s2(0)), ("S2/75", synth.scene_s2(75)),
              ("S2/150", synth.scene_s2(150))]
    tiles = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [6, 5, 4, 3]
    chunks = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
    for name, (depth, R, t) in scenes:
        d = torch.from_numpy(depth.view(np.int16)).cuda()
        dd = torch.empty_like(d)
        ctx.compute_dists(d.data_ptr(), synth.COLS * 2, dd.data_ptr(), synth.COLS * 2, synth.ROWS, synth.COLS, synth.INTR)
        a = (dd.data_ptr(), synth.COLS * 2, synth.ROWS, synth.COLS, R.reshape(-1), t, synth.INTR)
        for zc in chunks:
            ctx.set_option("z_chunk", zc)
            line = "%-7s zc=%-3d" % (name, zc)
            for tile in tiles:
                ctx.set_option("tile", tile)
                ctx.clear_volume(desc)
                for _ in range(3):
                    ctx.integrate(desc, *a)
                ctx.set_option("kernel_timing", 1)
                for _ in range(20):
                    ctx.integrate(desc, *a)
                ms, k = ctx.kernel_time()
                ctx.set_option("kernel_timing", 0)
                line += "  tile %d: %.4f ms" % (tile, ms / k)
            print(line, flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
