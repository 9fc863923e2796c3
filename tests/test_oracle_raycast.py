"""CPU tests of the ray-casting restatement (oracle/yak_oracle.c: yko_raycast).

The reference holds no test, fixture or golden vector for raycast (PARITY UNPINNED), and
its implementation is CUDA-only.  What can be pinned here: analytic known answers derived from the
cited source lines (a planar linear field, where trilinear interpolation and the secant step of
tsdf_volume.cu:333-338 are exact), physical consistency with the depth frames that were fused, and a
committed fixture that guards the oracle against drift.
"""
import os

import numpy as np

import oracle
from yak_amd import synth


def _cam2vol(R_v2c, t_v2c):
    R = np.ascontiguousarray(np.asarray(R_v2c, dtype=np.float32).T)
    t = (-(np.asarray(R_v2c, dtype=np.float64).T @ np.asarray(t_v2c, dtype=np.float64))).astype(np.float32)
    return R, t


def _plane_volume(n=32, k0=16, trunc_vox=4):
    """tsdf(x,y,z=k) = clamp((k0 - k) / trunc_vox, -1, 1): multiples of 1/4, exact in half; weight 1."""
    k = np.arange(n, dtype=np.float32)
    f = np.clip((k0 - k) / trunc_vox, -1, 1).astype(np.float32)
    words = oracle.float2half(f).astype(np.uint32) | (1 << 16)
    return np.ascontiguousarray(np.broadcast_to(words[:, None, None], (n, n, n))).reshape(-1)


def test_raycast_planar_field_known_answer():
    n, k0, voxel, trunc = 32, 16, 0.1, 0.4
    vol = _plane_volume(n, k0, 4)
    dims, vs = (n, n, n), (voxel,) * 3
    R = np.eye(3, dtype=np.float32)
    t = np.array([1.55, 1.55, -1.0], dtype=np.float32)       # camera 1 m in front of the z = 0 face
    intr = (100.0, 100.0, 7.5, 5.5)
    pts, nrm, fetches, hits = oracle.raycast(vol, dims, vs, trunc, R, t, R, intr, 12, 16)
    assert hits == 12 * 16 and not np.isnan(pts).any()
    # zero crossing of the field is the plane z_vol = k0 * voxel = 1.6 m => z_cam = 2.6 m for every
    # pixel; the secant step between two samples of a linear field lands on it up to fp32 rounding
    assert np.abs(pts[..., 2] - 2.6).max() < 2e-5
    # pinhole: x_cam = z * (u - cx) / fx
    u = np.arange(16, dtype=np.float64)[None, :]
    v = np.arange(12, dtype=np.float64)[:, None]
    assert np.abs(pts[..., 0] - 2.6 * (u - 7.5) / 100.0).max() < 2e-5
    assert np.abs(pts[..., 1] - 2.6 * (v - 5.5) / 100.0).max() < 2e-5
    assert np.all(pts[..., 3] == 0) and np.all(nrm[..., 3] == 0)
    # the field decreases along +z only: normal (0, 0, -1) up to the rounding of the trilinear weights
    assert np.abs(nrm[..., :3] - np.float32([0, 0, -1])).max() < 1e-6
    # the march starts at the near face and samples every trunc * 0.75 = 0.3 m
    assert fetches >= hits * 5

    dep, nrm2, _, hits2 = oracle.raycast(vol, dims, vs, trunc, R, t, R, intr, 12, 16, want_depth=True)
    assert hits2 == hits and np.array_equal(nrm2.view(np.uint32), nrm.view(np.uint32))
    assert np.array_equal(dep, (pts[..., 2] * np.float32(1000)).astype(np.uint16))   # truncation, cu:280
    assert set(np.unique(dep)) <= {2599, 2600}


def test_raycast_rotated_camera_reports_camera_frame():
    # the same plane seen by a camera yawed 20 degrees: vertices come back in the CAMERA frame
    n, voxel, trunc = 32, 0.1, 0.4
    vol = _plane_volume(n, 16, 4)
    a = np.deg2rad(20.0)
    R_c2v = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    eye = np.array([1.9, 1.55, -0.5])
    R_v2c, t_v2c = synth.vol2cam(R_c2v, eye)
    R, t = _cam2vol(R_v2c, t_v2c)
    pts, nrm, _, hits = oracle.raycast(vol, (n,) * 3, (voxel,) * 3, trunc, R, t, R_v2c, (100.0, 100.0, 7.5, 5.5), 12, 16)
    assert hits > 100
    m = ~np.isnan(pts[..., 0])
    back = pts[m][:, :3].astype(np.float64) @ R_c2v.T + eye      # camera -> volume
    assert np.abs(back[:, 2] - 1.6).max() < 1e-4
    n_vol = nrm[m][:, :3].astype(np.float64) @ R_c2v.T
    assert np.abs(n_vol - np.array([0, 0, -1.0])).max() < 1e-5


def test_raycast_misses():
    n = 16
    dims, vs = (n,) * 3, (0.1,) * 3
    R = np.eye(3, dtype=np.float32)
    intr = (50.0, 50.0, 3.5, 2.5)
    empty = np.zeros(n ** 3, dtype=np.uint32)                     # cleared volume: tsdf 0 everywhere
    pts, nrm, fetches, hits = oracle.raycast(empty, dims, vs, 0.4, R, np.float32([0.8, 0.8, -1]), R, intr, 6, 8)
    assert hits == 0 and np.isnan(pts).all() and np.isnan(nrm).all() and fetches > 0
    dep, _, _, _ = oracle.raycast(empty, dims, vs, 0.4, R, np.float32([0.8, 0.8, -1]), R, intr, 6, 8, want_depth=True)
    assert not dep.any()
    # looking away from the box: tmin >= tmax for every ray, nothing is fetched (cu:313-314)
    back = np.diag([-1.0, 1.0, -1.0]).astype(np.float32)
    vol = _plane_volume(n, 8, 4)
    pts, _, fetches, hits = oracle.raycast(vol, dims, vs, 0.4, back, np.float32([0.8, 0.8, -1]), back, intr, 6, 8)
    assert hits == 0 and fetches == 0 and np.isnan(pts).all()
    # a - -> + transition is seen first (camera behind the surface): the march stops without a hit (cu:328-329)
    pts, _, fetches, hits = oracle.raycast(vol, dims, vs, 0.4, back, np.float32([0.8, 0.8, 2.4]), back, intr, 6, 8)
    assert hits == 0 and fetches > 0


def test_raycast_agrees_with_fused_depth_s1():
    """Fuse the S1 frame, ray-cast from the same pose: the sphere comes back where the depth frame had it."""
    n = 96
    dims, vs, tr = synth.volume_geometry(n)
    depth, R_v2c, t_v2c = synth.scene_s1()
    dists = oracle.compute_dists(depth, synth.INTR)
    vol = np.zeros(n ** 3, dtype=np.uint32)
    for _ in range(2):
        oracle.integrate(vol, dims, vs, tr, synth.MAX_WEIGHT, dists, R_v2c, t_v2c, synth.INTR, nthreads=4)
    R, t = _cam2vol(R_v2c, t_v2c)
    pts, nrm, _, hits = oracle.raycast(vol, dims, vs, tr, R, t, R_v2c, synth.INTR, synth.ROWS, synth.COLS, nthreads=4)
    assert hits > 30000
    z = pts[..., 2]
    m = ~np.isnan(z)
    # central part of the sphere (incidence below ~30 degrees; the silhouette is at ~113 px): every
    # ray hits, within one voxel of the fused depth
    u, v = np.meshgrid(np.arange(synth.COLS), np.arange(synth.ROWS))
    r_px = np.hypot(u - synth.INTR[2], v - synth.INTR[3])
    assert m[r_px < 60].all()
    inner = m & (r_px < 60)
    assert np.abs(z[inner] - depth[inner] / 1000.0).max() < vs[0]
    nn = nrm[inner][:, :3]
    assert np.abs(np.linalg.norm(nn, axis=1) - 1).max() < 1e-5
    assert (nn[:, 2] < 0).all()                                    # facing the camera
    # threads do not change anything
    p1, n1, _, h1 = oracle.raycast(vol, dims, vs, tr, R, t, R_v2c, synth.INTR, synth.ROWS, synth.COLS, nthreads=1)
    assert h1 == hits and np.array_equal(p1.view(np.uint32), pts.view(np.uint32))
    assert np.array_equal(n1.view(np.uint32), nrm.view(np.uint32))


def test_golden_fixture_raycast_s1_32():
    """tests/golden/raycast_s1_32.npz (tests/golden/make_golden.py, written by this oracle)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "raycast_s1_32.npz"))
    v = np.load(os.path.join(os.path.dirname(__file__), "golden", "s1_32.npz"))
    dims, vs, tr = synth.volume_geometry(32)
    pts, nrm, fetches, hits = oracle.raycast(np.ascontiguousarray(v["vol_f3"]), dims, vs, tr, g["R"], g["t"], g["Rinv"],
                                             g["intr"], 60, 80)
    assert fetches == int(g["fetches"]) and hits == int(g["hits"])
    assert np.array_equal(pts.view(np.uint32), g["points"])
    assert np.array_equal(nrm.view(np.uint32), g["normals"])


def fused_orbit_volume(n, frames):
    dims, vs, tr = synth.volume_geometry(n)
    vol = np.zeros(int(np.prod(dims)), dtype=np.uint32)
    for f in frames:
        depth, R, t = synth.scene_s2(f)
        oracle.integrate(vol, dims, vs, tr, synth.MAX_WEIGHT, oracle.compute_dists(depth, synth.INTR), R, t, synth.INTR,
                         nthreads=8)
    return vol, dims, vs, tr


def composite_min_key(parts):
    """Reference composite of per-slab results [(points, normals, keys)]: smallest key owns the pixel."""
    keys = np.stack([k for _, _, k in parts])
    best = keys.min(axis=0)
    assert ((keys == best).sum(axis=0)[best != 0x7FFFFFFF] == 1).all()          # keys of different slabs never tie
    win = keys.argmin(axis=0)
    pts = np.take_along_axis(np.stack([p for p, _, _ in parts]), win[None, ..., None], axis=0)[0]
    nrm = np.take_along_axis(np.stack([n for _, n, _ in parts]), win[None, ..., None], axis=0)[0]
    return pts, nrm, best


def test_slab_marches_composite_to_the_whole_march():
    """SURVEY.md 8e option 1 on the CPU: every slab marches every ray and examines the pairs it owns;
    the first march-ending pair over all slabs is the one the whole-volume march stops at."""
    vol, dims, vs, tr = fused_orbit_volume(48, [0, 60, 120, 180, 240])
    rows, cols = 60, 80
    intr = (72.5, 72.5, 39.5, 29.5)
    later = False
    for frame in (20, 100, 215):
        _, R_v2c, t_v2c = synth.scene_s2(frame)
        R, t = _cam2vol(R_v2c, t_v2c)
        Rinv = np.ascontiguousarray(R.T)
        pts, nrm, _, hits = oracle.raycast(vol, dims, vs, tr, R, t, Rinv, intr, rows, cols, nthreads=4)
        assert hits > 500
        whole_p, whole_n, whole_k = oracle.raycast_slab(vol, dims, vs, tr, R, t, Rinv, intr, rows, cols, 0, dims[2], nthreads=4)
        assert np.array_equal(whole_p.view(np.uint32), pts.view(np.uint32))
        assert (whole_k[~np.isnan(pts[..., 0])] != 0x7FFFFFFF).all() and (whole_k % 2 == (whole_k == 0x7FFFFFFF)).all()
        for cuts in ([0, 24, 48], [0, 7, 19, 20, 41, 48], [0, 1, 47, 48]):
            parts = [oracle.raycast_slab(vol, dims, vs, tr, R, t, Rinv, intr, rows, cols, a, b, nthreads=4)
                     for a, b in zip(cuts, cuts[1:])]
            cp, cn, ck = composite_min_key(parts)
            assert np.array_equal(ck, whole_k)
            assert np.array_equal(cp.view(np.uint32), pts.view(np.uint32))
            assert np.array_equal(cn.view(np.uint32), nrm.view(np.uint32))
            # a slab on its own sees later pairs too (the march went on where another slab's pair had ended it)
            later = later or any((k[(k != 0x7FFFFFFF)] > whole_k[(k != 0x7FFFFFFF)]).any() for _, _, k in parts)
    assert later
