#!/usr/bin/env python3
"""What the two-kernel integrate launch costs before any part is walked: a frame without returns (every brick dead) at 512^3
with defer_mixed=2.  Run under `rocprofv3 --kernel-trace --stats` for the two kernels' durations."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    from yak_amd import capi, synth

    ctx = capi.Context(0)
    ctx.set_option("defer_mixed", 2)
    n = 512
    rows, cols, intr = synth.ROWS, synth.COLS, synth.INTR
    dims, vs, tr = synth.volume_geometry(n)
    vol = torch.zeros(n ** 3, dtype=torch.int32, device="cuda")
    desc = capi.volume_desc(vol.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT)
    _, R, t = synth.scene_s2(0)
    dd = torch.zeros((rows, cols), dtype=torch.int16, device="cuda")
    for _ in range(5):
        ctx.integrate(desc, dd.data_ptr(), cols * 2, rows, cols, R.reshape(-1), t, intr)
    torch.cuda.synchronize()
    ctx.set_option("kernel_timing", 1)
    for _ in range(20):
        ctx.integrate(desc, dd.data_ptr(), cols * 2, rows, cols, R.reshape(-1), t, intr)
        torch.cuda.synchronize()
    ms, k = ctx.kernel_time()
    print("blank frame, two kernels: %.4f ms per launch (event pair around both kernels), launches %s" % (ms / k, ctx.launch_counts()))
    ctx.close()


if __name__ == "__main__":
    main()
