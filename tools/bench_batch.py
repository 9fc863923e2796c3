#!/usr/bin/env python3
"""Times yakhip_integrate_batch against K single-frame calls at 512^3 (GPU only; no oracle)."""
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
    scenes = {"S0 x4": [synth.scene_s0()] * 4, "S1 x4": [synth.scene_s1()] * 4,
              "S2 rig (0/75/150/225)": [synth.scene_s2(f) for f in (0, 75, 150, 225)]}
    for name, fr in scenes.items():
        args, keep = [], []
        for depth, R, t in fr:
            d = torch.from_numpy(depth.view(np.int16)).cuda()
            dd = torch.empty_like(d)
            ctx.compute_dists(d.data_ptr(), synth.COLS * 2, dd.data_ptr(), synth.COLS * 2, synth.ROWS, synth.COLS, synth.INTR)
            keep.append(dd)
            args.append((dd.data_ptr(), synth.COLS * 2, synth.ROWS, synth.COLS, R.reshape(-1), t, synth.INTR))
        ctx.clear_volume(desc)
        for k in (1, 2, 3, 4):
            sub = args[:k]
            for _ in range(3):
                ctx.integrate_batch(desc, sub)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ctx.integrate_batch(desc, sub)
            e1.record()
            torch.cuda.synchronize()
            tb = e0.elapsed_time(e1) / 20
            e0.record()
            for _ in range(20):
                for a in sub:
                    ctx.integrate(desc, *a)
            e1.record()
            torch.cuda.synchronize()
            ts = e0.elapsed_time(e1) / 20
            print("%-24s K=%d  batch %.4f ms (%.4f per frame)   sequential %.4f ms   speed-up %.2fx" % (
                name, k, tb, tb / k, ts, ts / tb), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
