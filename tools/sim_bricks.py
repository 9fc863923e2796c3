#!/usr/bin/env python3
"""CPU model of the integrate kernel's culling on an orbit frame: classifies every voxel of the volume
(dead / front with tsdf 1 / truncation band) in numpy and counts, for a given wave footprint and chunking, how many
wave-layers could be settled whole.  Used to choose brick shapes before spending GPU time.
Statistics only (float64, ideal positions): no parity claim rests on it.

  python tools/sim_bricks.py --view 3 --n 512
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yak_amd import synth  # noqa: E402


def voxel_classes(n, frame, noise_mm=1.0):
    """uint8 [z, y, x]: 0 dead, 1 front (update, tsdf 1), 2 band (update, tsdf < 1)"""
    dims, vs, tr = synth.volume_geometry(n)
    depth, R, t = synth.scene_s2(frame, noise_mm=noise_mm)
    fx, fy, cx, cy = synth.INTR
    rows, cols = depth.shape
    u, v = np.meshgrid(np.arange(cols), np.arange(rows))
    lam = np.sqrt(((u - cx) / fx) ** 2 + ((v - cy) / fy) ** 2 + 1.0)
    dists = (depth.astype(np.float64) * lam * 0.001).astype(np.float16).astype(np.float32)
    R = R.astype(np.float64)
    t = t.astype(np.float64)
    out = np.zeros((n, n, n), dtype=np.uint8)
    x = np.arange(n, dtype=np.float64) * vs[0]
    y = np.arange(n, dtype=np.float64) * vs[1]
    X, Y = np.meshgrid(x, y)  # [y, x]
    for z in range(n):
        zz = z * vs[2]
        px = R[0, 0] * X + R[0, 1] * Y + R[0, 2] * zz + t[0]
        py = R[1, 0] * X + R[1, 1] * Y + R[1, 2] * zz + t[1]
        pz = R[2, 0] * X + R[2, 1] * Y + R[2, 2] * zz + t[2]
        ok = pz > 0
        with np.errstate(divide="ignore", invalid="ignore"):
            cu = fx * px / pz + cx
            cv = fy * py / pz + cy
        ok &= (cu >= 0) & (cv >= 0) & (cu < cols) & (cv < rows)
        iu = np.where(ok, np.floor(cu), 0).astype(np.int64)
        iv = np.where(ok, np.floor(cv), 0).astype(np.int64)
        Dp = dists[iv, iu]
        ok &= Dp != 0
        sdf = Dp - np.sqrt(px * px + py * py + pz * pz)
        upd = ok & (sdf >= -tr)
        out[z] = np.where(upd, np.where(sdf >= tr, 1, 2), 0)
    return out


def tile_stats(cls, tx, ty, tz):
    """per tile of tx x ty x tz voxels: (any dead, any front, any band) as three bool arrays [Z/tz, Y/ty, X/tx]"""
    n = cls.shape[0]
    c = cls.reshape(n // tz, tz, n // ty, ty, n // tx, tx)
    res = []
    for k in range(3):
        res.append((c == k).any(axis=(1, 3, 5)))
    return res


def chunk_bounds(n, ramp=6, lo=8, hi=24):
    z, out = 0, []
    while z < n:
        ln = (z // ramp + 1) & ~1
        ln = max(lo, min(hi, ln))
        ln = min(ln, n - z)
        out.append((z, ln))
        z += ln
    return out


def model(cls, fx_, fy_, chunks, sub=0, verbose=True, cost=None):
    """Wave footprint fx_ x fy_ voxels per layer; bricks = footprint x chunk.  sub > 0: mixed bricks are split in z into
    pieces of `sub` layers that are classified again.  Returns instruction estimate (wave-level VALU, millions)."""
    cost = cost or dict(classify=60, replay=6, front=34, grp=60, detail=110, band=40, general=80, dead_layer=6, subcls=25)
    n = cls.shape[0]
    d1, f1, b1 = tile_stats(cls, fx_, fy_, 1)  # per wave-layer
    tot = dict(bricks=0, dead=0, front=0, mixed=0, wl_pass=0, wl_dead=0, wl_nob=0, wl_band=0, replay=0, wl_sub_dead=0, wl_sub_front=0)
    for zs, ln in chunks:
        sl = slice(zs, zs + ln)
        d, f, b = d1[sl], f1[sl], b1[sl]
        upd = f | b
        brick_dead = ~upd.any(axis=0)
        brick_front = ~(d | b).any(axis=0)
        mixed = ~(brick_dead | brick_front)
        tot["bricks"] += mixed.size
        tot["dead"] += int(brick_dead.sum())
        tot["front"] += int(brick_front.sum())
        tot["mixed"] += int(mixed.sum())
        tot["replay"] += int(mixed.sum()) * zs
        m = mixed[None, :, :]
        wl_pass = (~(d | b)) & m
        wl_dead = (~upd) & m
        if sub > 0:
            # layers in sub-pieces that are wholly dead / front are settled without the per-voxel path
            k = ln // sub
            if k > 0:
                dd = wl_dead[: k * sub].reshape(k, sub, *wl_dead.shape[1:]).all(axis=1)
                ff = wl_pass[: k * sub].reshape(k, sub, *wl_pass.shape[1:]).all(axis=1)
                tot["wl_sub_dead"] += int(dd.sum()) * sub
                tot["wl_sub_front"] += int(ff.sum()) * sub
                wl_dead = wl_dead.copy()
                wl_pass = wl_pass.copy()
                wl_dead[: k * sub] &= ~np.repeat(dd, sub, axis=0)
                wl_pass[: k * sub] &= ~np.repeat(ff, sub, axis=0)
        tot["wl_pass"] += int(wl_pass.sum())
        tot["wl_dead"] += int(wl_dead.sum())
        tot["wl_nob"] += int((upd & d & ~b & m).sum())
        tot["wl_band"] += int((b & m).sum())
    waves_per_brick = 1
    c = cost
    instr = (tot["bricks"] * c["classify"] + tot["replay"] * c["replay"] + tot["front"] * 0 +
             tot["wl_pass"] * c["grp"] + tot["wl_dead"] * c["detail"] + tot["wl_nob"] * (c["detail"] + 30) +
             tot["wl_band"] * (c["detail"] + c["band"] + c["general"]) +
             tot["wl_sub_dead"] * c["dead_layer"] + tot["wl_sub_front"] * c["front"])
    # front bricks: count their layers
    if verbose:
        wl_mixed = tot["wl_pass"] + tot["wl_dead"] + tot["wl_nob"] + tot["wl_band"] + tot["wl_sub_dead"] + tot["wl_sub_front"]
        print("footprint %3dx%-3d sub %d: bricks %d dead %.1f%% front %.1f%% mixed %.1f%% | wave-layers in mixed %d: pass %.1f%% "
              "all-dead %.1f%% no-band %.1f%% band %.1f%% sub-dead %.1f%% sub-front %.1f%% | replay steps %.2fM | est %.1fM instr"
              % (fx_, fy_, sub, tot["bricks"], 100.0 * tot["dead"] / tot["bricks"], 100.0 * tot["front"] / tot["bricks"],
                 100.0 * tot["mixed"] / tot["bricks"], wl_mixed, 100.0 * tot["wl_pass"] / wl_mixed, 100.0 * tot["wl_dead"] / wl_mixed,
                 100.0 * tot["wl_nob"] / wl_mixed, 100.0 * tot["wl_band"] / wl_mixed, 100.0 * tot["wl_sub_dead"] / wl_mixed,
                 100.0 * tot["wl_sub_front"] / wl_mixed, tot["replay"] / 1e6, instr / 1e6))
    return tot, instr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--view", type=int, default=3)
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--cache", default="/tmp/sim_bricks_%d_%d.npy")
    args = ap.parse_args()
    frame = list(range(0, 300, 38))[args.view]
    path = args.cache % (args.n, frame)
    if os.path.exists(path):
        cls = np.load(path)
    else:
        cls = voxel_classes(args.n, frame)
        np.save(path, cls)
    n = args.n
    tot = cls.size
    print("frame %d: dead %.1f%% front %.1f%% band %.2f%%" % (frame, 100.0 * (cls == 0).sum() / tot, 100.0 * (cls == 1).sum() / tot,
                                                             100.0 * (cls == 2).sum() / tot))
    chunks = chunk_bounds(n)
    for fx_, fy_ in ((256, 1), (64, 4), (32, 8), (16, 16)):
        for sub in (0, 2, 4):
            model(cls, fx_, fy_, chunks, sub)
    for lo, hi in ((8, 8), (16, 16), (4, 12)):
        print("chunks %d..%d" % (lo, hi))
        for fx_, fy_ in ((64, 4), (32, 8), (16, 16)):
            model(cls, fx_, fy_, chunk_bounds(n, lo=lo, hi=hi), 0)


if __name__ == "__main__":
    main()


# ---- the kernel's conservative brick test (classify_brick) restated on the CPU, to see what it leaves on the table ----
def conservative_classes(n, frame, fx_, fy_, chunks, cell=16, noise_mm=1.0, eps_scale=1.0):
    """returns array [chunk][Y/fy, X/fx] of classes 0 mixed, 1 dead, 2 front as classify_brick decides them (float64)"""
    dims, vs, tr = synth.volume_geometry(n)
    depth, R, t = synth.scene_s2(frame, noise_mm=noise_mm)
    fx, fy, cx, cy = synth.INTR
    rows, cols = depth.shape
    u, v = np.meshgrid(np.arange(cols), np.arange(rows))
    lam = np.sqrt(((u - cx) / fx) ** 2 + ((v - cy) / fy) ** 2 + 1.0)
    Dp = (depth.astype(np.float64) * lam * 0.001).astype(np.float16).astype(np.float64)
    trh = tr * (1 + 2.0 ** -9)
    a = Dp * (1 - 2.0 ** -20) - trh
    A2 = np.where((Dp != 0) & (a > 0), a * a * (1 - 2.0 ** -16), -1.0)
    b = Dp * (1 + 2.0 ** -20) + trh
    B2 = np.where(Dp != 0, b * b * (1 + 2.0 ** -16) * (1 + 2.0 ** -8), -1.0)
    ccw, cch = (cols + cell - 1) // cell, (rows + cell - 1) // cell
    pad = np.full((cch * cell, ccw * cell), np.inf)
    pad[:rows, :cols] = A2
    cmin = pad.reshape(cch, cell, ccw, cell).min(axis=(1, 3))
    pad = np.full((cch * cell, ccw * cell), -np.inf)
    pad[:rows, :cols] = B2
    cmax = pad.reshape(cch, cell, ccw, cell).max(axis=(1, 3))
    # 2-D range queries by brute force per brick (small counts): use summed tables of min/max via sparse tables instead
    R = R.astype(np.float64)
    t = t.astype(np.float64)
    P = max(abs(R[i, 0]) * vs[0] * n + abs(R[i, 1]) * vs[1] * n + abs(R[i, 2]) * vs[2] * n + abs(t[i]) for i in range(3))
    e = (n + 32) * 2.0 ** -22 * P * eps_scale
    f = max(fx, fy)
    zmin_ok = max(max(f * P * 2.0 ** -15, 2.0 ** -10), 8 * e)
    out = []
    bx = np.arange(0, n, fx_)
    by = np.arange(0, n, fy_)
    for zs, ln in chunks:
        cls = np.zeros((len(by), len(bx)), dtype=np.uint8)
        corners = []
        for cz in (zs, zs + ln - 1):
            for cyv in (0, fy_ - 1):
                for cxv in (0, fx_ - 1):
                    X, Y = np.meshgrid((bx + cxv) * vs[0], (by + cyv) * vs[1])
                    Z = cz * vs[2]
                    corners.append(np.stack([R[i, 0] * X + R[i, 1] * Y + R[i, 2] * Z + t[i] for i in range(3)]))
        c = np.stack(corners)  # [8, 3, by, bx]
        xmin, xmax = c[:, 0].min(0) - e, c[:, 0].max(0) + e
        ymin, ymax = c[:, 1].min(0) - e, c[:, 1].max(0) + e
        zmin, zmax = c[:, 2].min(0) - e, c[:, 2].max(0) + e
        ax, ay, az = np.abs(c[:, 0]) + e, np.abs(c[:, 1]) + e, np.abs(c[:, 2]) + e
        d2max = (ax * ax + ay * ay + az * az).max(0)
        with np.errstate(divide="ignore", invalid="ignore"):
            uu = fx * c[:, 0] / c[:, 2] + cx
            vv = fy * c[:, 1] / c[:, 2] + cy
        umin, umax, vmin, vmax = uu.min(0), uu.max(0), vv.min(0), vv.max(0)
        for j in range(len(by)):
            for i in range(len(bx)):
                if zmax[j, i] <= 0:
                    cls[j, i] = 1
                    continue
                if zmin[j, i] < zmin_ok:
                    continue
                rz = 1.0 / (zmin[j, i] - e)
                du = fx * e * (1 + max(abs(xmin[j, i]), abs(xmax[j, i])) * rz) * rz * 1.01 + 1 / 32
                dv = fy * e * (1 + max(abs(ymin[j, i]), abs(ymax[j, i])) * rz) * rz * 1.01 + 1 / 32
                u0, u1, v0, v1 = umin[j, i] - du, umax[j, i] + du, vmin[j, i] - dv, vmax[j, i] + dv
                if u1 < 0 or v1 < 0 or u0 >= cols or v0 >= rows:
                    cls[j, i] = 1
                    continue
                px0, px1 = int(max(np.floor(u0), 0)), int(min(np.floor(u1), cols - 1))
                py0, py1 = int(max(np.floor(v0), 0)), int(min(np.floor(v1), rows - 1))
                sub_min = cmin[py0 // cell:py1 // cell + 1, px0 // cell:px1 // cell + 1]
                sub_max = cmax[py0 // cell:py1 // cell + 1, px0 // cell:px1 // cell + 1]
                if sub_min.size > 256:
                    continue
                mx = xmin[j, i] if xmin[j, i] > 0 else (-xmax[j, i] if xmax[j, i] < 0 else 0.0)
                my = ymin[j, i] if ymin[j, i] > 0 else (-ymax[j, i] if ymax[j, i] < 0 else 0.0)
                d2min = (mx * mx + my * my + zmin[j, i] ** 2) * (1 - 2.0 ** -11)
                inside = u0 >= 0.5 and v0 >= 0.5 and u1 < cols - 0.5 and v1 < rows - 0.5
                if d2min > sub_max.max():
                    cls[j, i] = 1
                elif inside and d2max[j, i] * (1 + 2.0 ** -11) <= sub_min.min():
                    cls[j, i] = 2
        out.append(cls)
    return out


def compare_conservative(n, frame, cls_vox, fx_, fy_, chunks, cell=16, eps_scale=1.0, label=""):
    cons = conservative_classes(n, frame, fx_, fy_, chunks, cell=cell, eps_scale=eps_scale)
    d1, f1, b1 = tile_stats(cls_vox, fx_, fy_, 1)
    tot = dict(n=0, dead=0, front=0, mixed=0, ideal_dead=0, ideal_front=0, wl_mixed=0, wl_ideal_mixed=0)
    for (zs, ln), cc in zip(chunks, cons):
        sl = slice(zs, zs + ln)
        upd = (f1[sl] | b1[sl]).any(axis=0)
        notfront = (d1[sl] | b1[sl]).any(axis=0)
        tot["n"] += cc.size
        tot["dead"] += int((cc == 1).sum())
        tot["front"] += int((cc == 2).sum())
        tot["mixed"] += int((cc == 0).sum())
        tot["ideal_dead"] += int((~upd).sum())
        tot["ideal_front"] += int((~notfront).sum())
        tot["wl_mixed"] += int((cc == 0).sum()) * ln
        tot["wl_ideal_mixed"] += int((upd & notfront).sum()) * ln
        assert not ((cc == 1) & upd).any(), "conservative test called an updating brick dead"
        assert not ((cc == 2) & notfront).any(), "conservative test called a brick front that is not"
    print("%s footprint %dx%d cell %d eps x%g: dead %.1f%% (ideal %.1f) front %.1f%% (ideal %.1f) mixed %.1f%% (ideal %.1f) | mixed wave-layers %d (ideal %d)"
          % (label, fx_, fy_, cell, eps_scale, 100.0 * tot["dead"] / tot["n"], 100.0 * tot["ideal_dead"] / tot["n"], 100.0 * tot["front"] / tot["n"],
             100.0 * tot["ideal_front"] / tot["n"], 100.0 * tot["mixed"] / tot["n"],
             100.0 * (tot["n"] - tot["ideal_dead"] - tot["ideal_front"]) / tot["n"], tot["wl_mixed"], tot["wl_ideal_mixed"]))
    return cons
