import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yak_amd import capi, synth
ctx = capi.Context(0)
n = 512
dims, vs, tr = synth.volume_geometry(n)
vol = torch.zeros(n ** 3, dtype=torch.int32, device="cuda")
desc = capi.volume_desc(vol.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT)
scenes = {"S0x4": [synth.scene_s0()] * 4, "S2rig": [synth.scene_s2(f) for f in (0, 75, 150, 225)]}
for name, fr in scenes.items():
    args, keep = [], []
    for depth, R, t in fr:
        d = torch.from_numpy(depth.view(np.int16)).cuda(); dd = torch.empty_like(d)
        ctx.compute_dists(d.data_ptr(), 1280, dd.data_ptr(), 1280, 480, 640, synth.INTR)
        keep.append(dd); args.append((dd.data_ptr(), 1280, 480, 640, R.reshape(-1), t, synth.INTR))
    for zc in (32, 48, 64, 96, 128):
        ctx.set_option("z_chunk", zc)
        line = "%s zc=%d" % (name, zc)
        for k in (2, 4):
            for _ in range(3): ctx.integrate_batch(desc, args[:k])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): ctx.integrate_batch(desc, args[:k])
            e1.record(); torch.cuda.synchronize()
            line += "  K=%d %.4f ms/frame" % (k, e0.elapsed_time(e1) / 20 / k)
        print(line, flush=True)
