"""CPU tests of the tracking-side restatements in oracle/yak_oracle.c: bilateral filter, depth
pyramid, point/normal maps, their 2x2 resize, and one ICP linearisation.

The reference holds no test or vector for any of them (PARITY UNPINNED).  Pinned here
by hand-computed cases read off the cited source lines, by independent numpy restatements, and by a
physical known answer: Gauss-Newton on the oracle's normal equations recovers a known camera motion.
"""
import math

import numpy as np

import oracle
from yak_amd import synth


# ------------------------------------------------------------------------------------ bilateral
def _bilateral_numpy(src, ksz, sigma_spatial, sigma_depth):
    """imgproc.cu:10-42 restated independently (python loops; small images only)."""
    rows, cols = src.shape
    sd = np.float32(sigma_depth) * np.float32(1000)
    s2 = np.float32(0.5) / (np.float32(sigma_spatial) * np.float32(sigma_spatial))
    d2 = np.float32(0.5) / (sd * sd)
    out = np.zeros_like(src)
    for y in range(rows):
        for x in range(cols):
            value = int(src[y, x])
            tx, ty = min(x - ksz // 2 + ksz, cols - 1), min(y - ksz // 2 + ksz, rows - 1)
            sum1 = np.float32(0)
            sum2 = np.float32(0)
            for cy in range(max(y - ksz // 2, 0), ty):
                for cx in range(max(x - ksz // 2, 0), tx):
                    depth = int(src[cy, cx])
                    space2 = np.float32((x - cx) ** 2 + (y - cy) ** 2)
                    color2 = np.float32((value - depth) ** 2)
                    w = np.float32(math.exp(-float(np.float32(space2 * s2 + color2 * d2))))
                    sum1 = np.float32(sum1 + np.float32(depth) * w)
                    sum2 = np.float32(sum2 + w)
            with np.errstate(invalid="ignore", divide="ignore"):
                q = np.float32(sum1) / np.float32(sum2)
            out[y, x] = 0 if np.isnan(q) else int(np.rint(q))
    return out


def test_bilateral_constant_and_zero_images():
    const = np.full((12, 17), 1234, dtype=np.uint16)
    assert np.array_equal(oracle.bilateral_filter(const), const)
    assert not oracle.bilateral_filter(np.zeros((9, 9), dtype=np.uint16)).any()


def test_bilateral_matches_independent_restatement():
    depth, _, _ = synth.scene_s2(40)
    patch = np.ascontiguousarray(depth[200:224, 300:330])          # sphere rim + ground: edges and zeros
    got = oracle.bilateral_filter(patch, 7, 4.5, 0.04)
    want = _bilateral_numpy(patch, 7, 4.5, 0.04)
    diff = np.abs(got.astype(int) - want.astype(int))
    assert diff.max() <= 1 and (diff != 0).mean() < 0.01           # expf vs exp(double) at a rounding boundary
    # edge preservation: a 1 m step is not smeared (weights across it underflow)
    step = np.full((16, 16), 1000, dtype=np.uint16)
    step[:, 8:] = 2000
    f = oracle.bilateral_filter(step)
    assert np.array_equal(f[:, :15], step[:, :15])
    # the window's exclusive upper bound is clipped to size - 1 (imgproc.cu:20-21): in the last column the
    # window is the 3 columns to the left only -- here the far side of the step, so the 2 m value survives
    assert np.array_equal(f[:, 15], step[:, 15])


def test_truncate_depth():
    d = np.array([[0, 999, 1000, 1001, 65535]], dtype=np.uint16)
    assert oracle.truncate_depth(d, 1.0).tolist() == [[0, 999, 1000, 0, 0]]
    assert oracle.truncate_depth(d, 100.0).tolist() == d.tolist()    # 100 m saturates the u16 limit


# ------------------------------------------------------------------------------------ pyramid
def test_depth_pyramid_hand_cases():
    const = np.full((8, 12), 700, dtype=np.uint16)
    assert np.array_equal(oracle.depth_pyramid(const), np.full((4, 6), 700, dtype=np.uint16))
    # neighbours further than 3 * sigma_depth (120 mm) from the centre are left out of the mean
    img = np.full((8, 8), 1000, dtype=np.uint16)
    img[:, 4:] = 1500
    p = oracle.depth_pyramid(img)
    assert np.array_equal(p[:, :2], np.full((4, 2), 1000)) and np.array_equal(p[:, 2:], np.full((4, 2), 1500))
    # integer mean, truncating: centre 100, window {100, 101, 101, ...}
    img = np.full((6, 6), 101, dtype=np.uint16)
    img[2, 2] = 100
    p = oracle.depth_pyramid(img)
    # window of dst(1,1): rows/cols 0..4 clipped by min(2*1-2+5, 6-1) = 5 -> 5 x 5 = 25 taps: 24 * 101 + 100
    assert p[1, 1] == (24 * 101 + 100) // 25


# ------------------------------------------------------------------------------------ point / normal maps
def test_points_normals_fronto_parallel_plane():
    rows, cols = 10, 14
    depth = np.full((rows, cols), 2000, dtype=np.uint16)
    intr = (100.0, 110.0, 6.5, 4.5)
    pts, nrm = oracle.compute_points_normals(depth, intr)
    # last row / column: no forward neighbour -> NaN in all four lanes (imgproc.cu:219-222)
    assert np.isnan(pts[-1]).all() and np.isnan(pts[:, -1]).all() and np.isnan(nrm[-1]).all()
    inner_p, inner_n = pts[:-1, :-1], nrm[:-1, :-1]
    # (0, 0, -1) up to the rounding of x * (1 / sqrt(x * x)); x and y components are exactly (-)0
    assert (inner_n[..., :2] == 0).all() and (inner_n[..., 3] == 0).all()
    assert np.abs(inner_n[..., 2] + 1).max() <= 2 ** -23
    z = np.float32(2000) * np.float32(0.001)
    u = np.arange(cols - 1, dtype=np.float32)[None, :]
    v = np.arange(rows - 1, dtype=np.float32)[:, None]
    finvx, finvy = np.float32(1) / np.float32(100), np.float32(1) / np.float32(110)
    assert np.array_equal(inner_p[..., 0], np.broadcast_to(z * (u - np.float32(6.5)) * finvx, inner_p.shape[:2]))
    assert np.array_equal(inner_p[..., 1], np.broadcast_to(z * (v - np.float32(4.5)) * finvy, inner_p.shape[:2]))
    assert (inner_p[..., 2] == z).all() and (inner_p[..., 3] == 0).all()
    # a missing neighbour knocks out the pixels that use it (z00 * z01 * z10 != 0, :229)
    depth[4, 5] = 0
    pts, _ = oracle.compute_points_normals(depth, intr)
    assert np.isnan(pts[4, 5, 0]) and np.isnan(pts[4, 4, 0]) and np.isnan(pts[3, 5, 0]) and not np.isnan(pts[3, 4, 0])


def test_resize_points_normals():
    rng = np.random.default_rng(3)
    p = rng.normal(size=(6, 8, 4)).astype(np.float32)
    n = rng.normal(size=(6, 8, 4)).astype(np.float32)
    p[2, 3, 0] = np.nan                                              # knocks out output (1, 1)
    po, no = oracle.resize_points_normals(p, n)
    assert po.shape == (3, 4, 4)
    assert np.isnan(po[1, 1, :3]).all() and np.isnan(no[1, 1, :3]).all() and po[1, 1, 3] == 0 and no[1, 1, 3] == 0
    want = (p[0, 0, :3] + p[0, 1, :3] + p[1, 0, :3] + p[1, 1, :3]) * np.float32(0.25)
    assert np.array_equal(po[0, 0, :3], want) and po[0, 0, 3] == 0
    want = (n[4, 6, :3] + n[4, 7, :3] + n[5, 6, :3] + n[5, 7, :3]) * np.float32(0.25)
    assert np.array_equal(no[2, 3, :3], want)


# ------------------------------------------------------------------------------------ ICP
def _maps(frame, level=0):
    depth, R_v2c, t_v2c = synth.scene_s2(frame)
    d = oracle.bilateral_filter(depth)
    for _ in range(level):
        d = oracle.depth_pyramid(d)
    div = 1 << level
    intr = tuple(np.float32(v) / np.float32(div) for v in synth.INTR)
    pts, nrm = oracle.compute_points_normals(d, intr)
    return pts, nrm, R_v2c.astype(np.float64), t_v2c.astype(np.float64)


def _unpack(sums):
    """StreamHelper::get, projective_icp.cpp:42-61"""
    A = np.zeros((6, 6))
    b = np.zeros(6)
    k = 0
    for i in range(6):
        for j in range(i, 7):
            if j == 6:
                b[i] = sums[k]
            else:
                A[i, j] = A[j, i] = sums[k]
            k += 1
    return A, b


def _rodrigues(r):
    th = np.linalg.norm(r)
    if th < 1e-12:
        return np.eye(3)
    k = r / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * K @ K


def test_icp_identity_on_identical_maps():
    pts, nrm, _, _ = _maps(10, level=1)
    rows, cols = pts.shape[:2]
    sums, counts, codes = oracle.icp_accumulate(np.eye(3), np.zeros(3), synth.INTR, 1, 0.1, math.radians(30.0), pts, nrm,
                                                pts, nrm, want_codes=True)
    assert counts.sum() == rows * cols
    assert counts[1] == np.isnan(pts[..., 0]).sum() and np.array_equal(codes == 40, np.isnan(pts[..., 0]))
    assert counts[0] > 0.2 * rows * cols
    A, b = _unpack(sums)
    assert np.all(np.linalg.eigvalsh(A) > 0)
    # a point matched with itself has zero residual; the few that project into a neighbour pixel do not
    assert np.abs(np.linalg.solve(A, b)).max() < 1e-3
    assert sorted(np.unique(codes).tolist())[0] == 0 and set(np.unique(codes).tolist()) <= {0, 40, 80, 120, 160, 200}


def test_icp_gauss_newton_recovers_known_motion():
    """prev = frame 20, curr = frame 22 of the orbit (2.4 degrees apart).  Iterating the oracle's
    normal equations the way projective_icp.cpp:206-229 does converges to the true curr->prev motion."""
    vprev, nprev, Rp, tp = _maps(20, level=1)
    vcurr, ncurr, Rc, tc = _maps(22, level=1)
    # vol2cam poses: x_prev = Rp x_v + tp, x_curr = Rc x_v + tc  =>  x_prev = Rp Rc^T (x_curr - tc) + tp
    R_true = Rp @ Rc.T
    t_true = tp - R_true @ tc
    R, t = np.eye(3), np.zeros(3)
    for _ in range(12):
        sums, counts = oracle.icp_accumulate(R, t, synth.INTR, 1, 0.1, math.radians(30.0), vcurr, ncurr, vprev, nprev)
        A, b = _unpack(sums)
        assert abs(np.linalg.det(A)) > 1e-15 and counts[0] > 1000
        r = np.linalg.solve(A, b)
        Rinc = _rodrigues(r[:3])
        R, t = Rinc @ R, Rinc @ t + r[3:]
    ang = math.degrees(math.acos(min(1.0, (np.trace(R_true.T @ R) - 1) / 2)))
    assert ang < 0.1, ang                     # true rotation is 2.4 degrees
    assert np.linalg.norm(t - t_true) < 0.01, (t, t_true)   # true translation is ~0.13 m
