#!/usr/bin/env python3
"""Fused frames per second of the complete KinFu loop at 512^3 on the S2 orbit (yak_amd/loop.py).

  python tools/bench_loop.py [--n 512] [--frames 60] [--no-icp | --host-icp]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--no-icp", action="store_true")
    ap.add_argument("--host-icp", action="store_true", help="one blocking linearisation at a time, numpy solve")
    ap.add_argument("--no-overlap", action="store_true", help="pre-processing on the main stream, behind the frame before")
    ap.add_argument("--no-prepare", action="store_true", help="the integrate prepass stays on the main stream")
    ap.add_argument("--no-poll", action="store_true", help="yakhip_icp_estimate copies and synchronises instead of polling")
    args = ap.parse_args()
    from yak_amd import capi
    from yak_amd.loop import run_kinfu_loop

    ctx = capi.Context(0)
    if args.no_poll:
        ctx.set_option("icp_poll", 0)
    run_kinfu_loop(ctx, args.n, 8, "device", overlap=not args.no_overlap, prepare=not args.no_prepare)  # warm-up
    mode = None if args.no_icp else ("host" if args.host_icp else "device")
    c0 = ctx.launch_counts()
    r = run_kinfu_loop(ctx, args.n, args.frames, mode, overlap=not args.no_overlap, prepare=not args.no_prepare)
    c1 = ctx.launch_counts()
    print("integrate launches: %s" % {k: c1[k] - c0[k] for k in c1})
    print("%d^3, %d frames of the S2 orbit, ICP %s: %.1f frames/s (wall %.3f s); worst position error %.1f mm" % (
        r["n"], r["frames"], {None: "off (true poses)", "host": "on (host loop)", "device": "on (device loop)"}[mode],
        r["frames_per_s"], r["wall_s"], r["worst_position_error_mm"]))
    print("per frame, stream time between events [ms]: " + ", ".join("%s %.3f" % kv for kv in r["stage_ms"].items()))
    ms = r["icp_ms_per_frame"][1:]
    if mode == "device" and ms:
        med = sorted(ms)[len(ms) // 2]
        slow = [i + 1 for i, v in enumerate(ms) if v > 2.0 * med]
        print("ICP per frame: median %.3f ms, max %.3f ms; %d frames above twice the median (first call, host-solve path or a host stall): %s" % (
            med, max(ms), len(slow), ["%d: %.2f ms" % (i, ms[i - 1]) for i in slow[:40]]))
    ctx.close()


if __name__ == "__main__":
    main()
