#!/usr/bin/env python3
"""Times yakhip_raycast_points on a 512^3 volume fused from the S2 orbit scene (GPU only; no oracle).
Usage: python tools/bench_raycast.py [--n 512] [--reps 20]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def phase_cycles(capi):
    """Phase sums of a library built with EXTRA=-DYAK_RAYCAST_TIMING (None otherwise); resets them."""
    import ctypes
    try:
        fn = capi.load().yakhip_debug_raycast_timing
    except AttributeError:
        return None
    out = (ctypes.c_ulonglong * 10)()
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    return list(out) if fn(out, 1) == 0 else None


def wave_rows(capi):
    import ctypes
    try:
        fn = capi.load().yakhip_debug_raycast_rows
    except AttributeError:
        return None
    out = np.zeros((8192, 10), dtype=np.uint64)
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    n = fn(out.ctypes.data, 8192)
    return out[:n] if n > 0 else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--no-bands", action="store_true")
    ap.add_argument("--bands", type=int, default=None)
    ap.add_argument("--empty", action="store_true", help="ray cast the cleared volume")
    ap.add_argument("--segments", type=int, default=None, help="waves per ray with the surface map (1, 2, 4)")
    args = ap.parse_args()
    import torch

    from yak_amd import capi, synth

    ctx = capi.Context(0)
    if args.no_bands:
        ctx.set_option("raycast_bands", 0)
    if args.bands is not None:
        ctx.set_option("raycast_bands", args.bands)
    if args.segments is not None:
        ctx.set_option("raycast_segments", args.segments)
    n = args.n
    dims, vs, tr = synth.volume_geometry(n)
    vol = torch.zeros(n ** 3, dtype=torch.int32, device="cuda")
    plain = capi.volume_desc(vol.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT)
    smap = torch.zeros(capi.surface_map_bytes(plain) // 4, dtype=torch.int32, device="cuda")
    desc = capi.volume_desc(vol.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT, surface_map=smap.data_ptr())
    d_dists = torch.empty((synth.ROWS, synth.COLS), dtype=torch.int16, device="cuda")
    for f in ([] if args.empty else range(0, 300, 15)):
        depth, R, t = synth.scene_s2(f)
        d_depth = torch.from_numpy(depth.view(np.int16)).cuda()
        ctx.compute_dists(d_depth.data_ptr(), synth.COLS * 2, d_dists.data_ptr(), synth.COLS * 2, synth.ROWS, synth.COLS,
                          synth.INTR)
        ctx.integrate(desc, d_dists.data_ptr(), synth.COLS * 2, synth.ROWS, synth.COLS, R.reshape(-1), t, synth.INTR)
    torch.cuda.synchronize()
    nw = ((n + 15) // 16) ** 3 // 32  # brick words; the cell words follow
    marked = int(sum(bin(int(w) & 0xffffffff).count("1") for w in smap.cpu().numpy()[:nw]))
    kept = smap.clone()
    ctx.surface_map_rebuild(desc)
    torch.cuda.synchronize()
    exact = int(sum(bin(int(w) & 0xffffffff).count("1") for w in smap.cpu().numpy()[:nw]))
    missing = int((smap & ~kept).ne(0).sum().item())
    print("surface map: %d bricks marked by integrate, %d hold a negative voxel, %d of those unmarked (must be 0), %d bricks"
          % (marked, exact, missing, (n // 16) ** 3), flush=True)
    smap.copy_(kept)
    pts = torch.empty((synth.ROWS, synth.COLS, 4), dtype=torch.float32, device="cuda")
    nrm = torch.empty_like(pts)
    for view in (0, 37, 100, 150, 222):
        _, R_v2c, t_v2c = synth.scene_s2(view)
        R = np.ascontiguousarray(R_v2c.T)
        t = (-(R_v2c.astype(np.float64).T @ t_v2c.astype(np.float64))).astype(np.float32)
        results = []
        for name, d in (("no map", plain), ("map", desc)):
            call = lambda: ctx.raycast_points(d, R.reshape(-1), t, R_v2c.reshape(-1), synth.INTR, pts.data_ptr(),  # noqa: E731
                                              synth.COLS * 16, nrm.data_ptr(), synth.COLS * 16, synth.ROWS, synth.COLS)
            for _ in range(3):
                call()
            phase_cycles(capi)
            ctx.set_option("kernel_timing", 1)
            for _ in range(args.reps):
                call()
            ms, k = ctx.kernel_time()
            ctx.set_option("kernel_timing", 0)
            rows = wave_rows(capi)
            phases = phase_cycles(capi)
            hits = int((~torch.isnan(pts[..., 0])).sum().item())
            results.append((pts.clone(), nrm.clone()))
            print("view %3d %-6s: raycast_kernel %.4f ms (mean of %d), %d hits of %d rays" % (view, name, ms / k, k, hits,
                                                                                           synth.ROWS * synth.COLS), flush=True)
            if phases:
                w = max(phases[5], 1)
                print("   per wave (cycles): map fill %.0f, entry + cell walk %.0f, %.1f trips %.0f (of them %.1f with a fetch: %.0f), "
                      "hit %.0f; %.1f hits per wave; longest wave %d"
                      % (phases[0] / w, phases[1] / w, phases[4] / w, phases[2] / w, phases[8] / w, phases[9] / w, phases[3] / w,
                         phases[6] / w, phases[7]), flush=True)
                if rows is not None and name == "map":
                    live = rows[rows[:, 5] > 0]
                    order = np.argsort(-live[:, 7].astype(np.int64))
                    print("   waves by lifetime (cycles: total / cell walk / march / hit; trips, trips with a fetch, hits):")
                    for q in (0, 1, 2, len(order) // 100, len(order) // 10, len(order) // 2, len(order) * 9 // 10):
                        r = live[order[q]]
                        print("     #%-5d %7d / %6d / %7d / %6d; %3d %3d %2d" % (q, r[7], r[1], r[2], r[3], r[4], r[8], r[6]), flush=True)
        same = all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(results[0], results[1]))
        print("view %3d: maps with and without the surface map are %s" % (view, "bit-identical" if same else "DIFFERENT"), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
