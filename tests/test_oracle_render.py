"""CPU checks of the oracle's debug-display restatement (imgproc.cu:400-561): hand-computed pixels and an
independent numpy fp32 restatement.  PARITY UNPINNED: the reference holds no rendered image, and its
specular term uses the approximate __powf (the oracle multiplies; a grey level may differ by one)."""
import numpy as np

import oracle
from yak_amd import synth

f32 = np.float32


def maps(points, normals):
    p = np.full((1, len(points), 4), 0, dtype=np.float32)
    n = np.zeros_like(p)
    p[0, :, :3] = points
    n[0, :, :3] = normals
    return p, n


def test_phong_known_pixels():
    # eye and light at the origin, surface point one metre ahead
    P = [(0, 0, 1)] * 3
    N = [(0, 0, -1),   # facing the eye: 0.3 + 0.5 * 1 + 0.2 * 1^20 = 1 -> 255
         (0, 0, 1),    # facing away: no diffuse; R = 2 N (N.L) - L = -z still mirrors onto V -> 0.3 + 0.2 -> 127
         (1, 0, 0)]    # edge on: N.L = 0, R = -L points away from the eye -> ambient only, 0.3 * 255 = 76.5 -> 76
    img = oracle.render_image(*maps(P, N), light_pose=(0, 0, 0))
    assert img[0, :, 0].tolist() == [255, 127, 76]
    assert (img[..., 0] == img[..., 1]).all() and (img[..., 1] == img[..., 2]).all() and (img[..., 3] == 0).all()


def test_a_nan_normal_under_a_valid_point_keeps_the_ambient_term():
    # fmax(0, NaN) = 0 in both Phong terms (imgproc.cu:494): what is left is the ambient 0.3 -> 76
    img = oracle.render_image(*maps([(0, 0, 1), (0.1, 0, 2)], [(np.nan, np.nan, np.nan), (0, np.nan, -1)]), light_pose=(0, 0, 0))
    assert img[0, 0].tolist() == [76, 76, 76, 0] and img[0, 1].tolist() == [76, 76, 76, 0]
    # a point ON the light (L = 0 / 0) likewise
    img = oracle.render_image(*maps([(0.5, 0.5, 1)], [(0, 0, -1)]), light_pose=(0.5, 0.5, 1))
    assert img[0, 0].tolist() == [76, 76, 76, 0]


def test_backdrop_where_there_is_no_point():
    rows, cols = 480, 3
    p = np.full((rows, cols, 4), np.nan, dtype=np.float32)
    n = np.zeros_like(p)
    img = oracle.render_image(p, n, light_pose=(0, 0, 0))
    w = np.arange(rows, dtype=f32) / f32(rows)
    one = f32(1)
    b = (f32(4) / f32(255)) * (one - w) + (f32(236) / f32(255)) * w
    g = (f32(2) / f32(255)) * (one - w) + (f32(120) / f32(255)) * w
    assert b.dtype == np.float32
    want_b, want_g = (b * f32(255)).astype(np.uint8), (g * f32(255)).astype(np.uint8)
    for x in range(cols):
        assert np.array_equal(img[:, x, 0], want_b)
        assert np.array_equal(img[:, x, 1], want_g) and np.array_equal(img[:, x, 2], want_g)
    assert img[0, 0].tolist() == [want_b[0], want_g[0], want_g[0], 0] and 3 <= img[0, 0, 0] <= 4
    assert 234 <= img[-1, 0, 0] <= 236 and 118 <= img[-1, 0, 1] <= 120     # blue-ish at the bottom (b, g, r order)


def test_tangent_colours_known_pixels():
    _, n = maps([(0, 0, 0)] * 4, [(0, 0, -1), (1, 0, 0), (0, 1, 0), (np.nan, np.nan, np.nan)])
    img = oracle.render_tangent_colors(n)
    # {b, g, r, 0}: r = (5 - 3.5 nx) 25.5, g = (5 - 2.5 ny) 25.5, b = (5 - 3.5 nz) 25.5, truncated
    assert img[0, 0].tolist() == [216, 127, 127, 0]     # nz = -1: 8.5 * 25.5 = 216.75
    assert img[0, 1].tolist() == [127, 127, 38, 0]      # nx = 1: 1.5 * 25.5 = 38.25
    assert img[0, 2].tolist() == [127, 63, 127, 0]      # ny = 1: 2.5 * 25.5 = 63.75
    assert img[0, 3].tolist() == [0, 0, 0, 0]           # NaN converts to 0


def numpy_phong(p, n, light):
    """The expression tree of imgproc.cu:475-496 in numpy fp32 (every intermediate rounded to fp32)."""
    def dot(a, b):  # temp_utils.hpp:36-39: fma(x1, x2, fma(y1, y2, z1 * z2)); exact products in float64, rounded once each
        inner = (a[..., 1].astype(np.float64) * b[..., 1] + (a[..., 2] * b[..., 2]).astype(np.float64)).astype(f32)
        return (a[..., 0].astype(np.float64) * b[..., 0] + inner.astype(np.float64)).astype(f32)

    def normalized(v):
        return v * (f32(1) / np.sqrt(dot(v, v)))[..., None]

    L = normalized(np.asarray(light, dtype=f32) - p)
    V = normalized(f32(0) - p)
    nl = dot(n, L)
    R = normalized(f32(2) * n * nl[..., None] - L)
    rv = np.maximum(dot(R, V), f32(0))
    v2 = rv * rv
    v4 = v2 * v2
    v5 = v4 * rv
    v10 = v5 * v5
    Ix = f32(0.3) + f32(0.5) * np.maximum(nl, f32(0)) + f32(0.2) * (v10 * v10)
    return (np.clip(Ix, 0, 1) * f32(255)).astype(np.uint8)


def test_points_and_depth_variants_against_numpy():
    depth, _, _ = synth.scene_s2(40)
    depth = np.ascontiguousarray(depth[::4, ::4])          # 120 x 160
    intr = [v / 4 for v in synth.INTR]
    pts, nrm = oracle.compute_points_normals(depth, intr)
    light = (0.4, -0.2, -0.5)
    img = oracle.render_image(pts, nrm, light)
    valid = ~np.isnan(pts[..., 0])
    assert valid.sum() > 5000
    want = numpy_phong(pts[valid][:, :3], nrm[valid][:, :3], light)
    got = img[valid][:, 0]
    # the fma emulation above rounds the inner sum twice in rare cases: allow one grey level on < 0.1 %
    diff = np.abs(got.astype(int) - want.astype(int))
    assert diff.max() <= 1 and (diff != 0).mean() < 1e-3
    assert 76 <= got.min() and got.max() > 150
    back = oracle.render_image(np.full_like(pts, np.nan), nrm, light)
    assert np.array_equal(img[~valid], back[~valid])

    # the depth variant re-projects z = d * 0.001 through the same intrinsics: where the point map holds
    # exactly that re-projection (compute_points_normals, imgproc.cu:226-229 vs :422) the pixels agree
    img_d = oracle.render_image(depth, nrm, light, intr=intr)
    assert np.array_equal(img_d[valid], img[valid])
    assert np.array_equal(img_d[depth == 0], back[depth == 0])
