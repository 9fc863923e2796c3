#!/usr/bin/env python3
"""Conditioning of the ICP normal equations along the S2 orbit (host-side loop): per frame the smallest ratio of
lambda_min to OpenCV's singular-value cut-off 2 * FLT_EPSILON * trace, and of the two lower bounds the device
solve certifies with (icp.hip solve_update).  GPU only.
  python tools/icp_cond.py [--frames 300] [--show 19 237]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--show", type=int, nargs="*", default=[])
    args = ap.parse_args()
    from yak_amd import capi
    from yak_amd.loop import run_kinfu_loop

    rec = {}

    def on_system(f, l, A, b):
        w = np.linalg.eigvalsh(A)
        tr = float(np.trace(A))
        cut = 2.0 * 1.1920928955078125e-7 * tr
        det = float(np.prod(w))
        amgm = det / (tr / 5.0) ** 5
        tight = 1.0 / float(np.sum(1.0 / w)) if w[0] > 0 else 0.0
        rec.setdefault(f, []).append((l, w[0] / cut, amgm / cut, tight / cut, det, w))

    ctx = capi.Context(0)
    run_kinfu_loop(ctx, 512, args.frames, "host", on_system=on_system)
    worst = sorted(rec.items(), key=lambda kv: min(r[3] for r in kv[1]))[:12]
    print("frames with the smallest certified margin (tight bound / cut-off; the device path needs > 4):")
    for f, rs in worst:
        r = min(rs, key=lambda r: r[3])
        print("  frame %3d level %d: lambda_min/cut %.3g, AM-GM bound/cut %.3g, tight bound/cut %.3g, det %.3g" % (f, r[0], r[1], r[2], r[3], r[4]))
    for f in args.show:
        for r in rec.get(f, []):
            print("frame %d level %d: lambda_min/cut %.3g amgm/cut %.3g tight/cut %.3g det %.3g eig %s" % (f, r[0], r[1], r[2], r[3], r[4], np.array2string(r[5], precision=3)))
    ctx.close()


if __name__ == "__main__":
    main()
