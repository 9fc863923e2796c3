"""yakhip_raycast_points_pyramid: the ray cast that also writes the half- and quarter-size maps KinFu builds from its
result (kinfu.cpp:294-306, resizePointsNormals level by level: imgproc.cu:348-395).  Every map must carry the bits
of yakhip_raycast_points followed by yakhip_resize_points_normals per level -- both of which are checked against the
oracle in their own tests -- NaN pixels, padding columns and odd sizes included, with and without the surface map.
"""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
from test_raycast_gpu import cam2vol, fused_volume  # noqa: E402

import oracle  # noqa: E402
from yak_amd import capi, synth  # noqa: E402

pytestmark = pytest.mark.gpu


def _maps(rows, cols, pad):
    """(points, normals) images of rows x (cols + pad) float4, filled with 7 so that stray stores show."""
    return [torch.full((max(rows, 1), cols + pad, 4), 7.0, dtype=torch.float32, device="cuda") for _ in range(2)]


def _bits(img, cols):
    h = img.cpu().numpy()
    assert (h[:, cols:] == 7.0).all(), "padding columns written"
    return h[:, :cols].view(np.uint32).copy()


def _run(ctx, desc, R, t, Rinv, intr, rows, cols, pad, levels):
    """Returns ([level 0..levels-1 (points bits, normals bits)] by the pyramid call, the same by separate calls)."""
    sizes = [(rows >> l, cols >> l) for l in range(3)]
    pose = (np.asarray(R).reshape(-1), t, np.asarray(Rinv).reshape(-1), intr)
    one = [_maps(r, c, pad) for r, c in sizes]
    ctx.raycast_points(desc, *pose, one[0][0].data_ptr(), (cols + pad) * 16, one[0][1].data_ptr(), (cols + pad) * 16, rows, cols)
    for l in range(1, levels):
        (sr, sc), (dr, dc) = sizes[l - 1], sizes[l]
        ctx.resize_points_normals(one[l - 1][0].data_ptr(), (sc + pad) * 16, one[l - 1][1].data_ptr(), (sc + pad) * 16, sr, sc,
                                  one[l][0].data_ptr(), (dc + pad) * 16, one[l][1].data_ptr(), (dc + pad) * 16)
    pyr = [_maps(r, c, pad) for r, c in sizes]
    lv = lambda l: (pyr[l][0].data_ptr(), (sizes[l][1] + pad) * 16, pyr[l][1].data_ptr(), (sizes[l][1] + pad) * 16)  # noqa: E731
    ctx.raycast_points_pyramid(desc, *pose, pyr[0][0].data_ptr(), (cols + pad) * 16, pyr[0][1].data_ptr(), (cols + pad) * 16,
                               rows, cols, lv(1), lv(2) if levels == 3 else None)
    torch.cuda.synchronize()
    take = lambda maps: [(_bits(maps[l][0], sizes[l][1]), _bits(maps[l][1], sizes[l][1])) for l in range(levels)]  # noqa: E731
    if levels == 2:  # the quarter-size maps were not asked for: untouched
        assert (pyr[2][0] == 7.0).all() and (pyr[2][1] == 7.0).all()
    return take(pyr), take(one)


def _check(got, want):
    for l, ((gp, gn), (wp, wn)) in enumerate(zip(got, want)):
        assert gp.shape == wp.shape
        assert np.array_equal(gp, wp), "level %d: %d point words differ" % (l, np.count_nonzero(gp != wp))
        assert np.array_equal(gn, wn), "level %d: %d normal words differ" % (l, np.count_nonzero(gn != wn))


@pytest.mark.parametrize("with_map", [False, True])
@pytest.mark.parametrize("view", [5, 133])
def test_pyramid_equals_raycast_then_resize_on_the_orbit(gpu_ctx, view, with_map):
    vol_h, dims, vs, tr = fused_volume(128, [0, 10, 20, 30, 40, 50, 120, 250], scene="s2")
    vol = torch.from_numpy(vol_h.view(np.int32)).cuda()
    plain = capi.volume_desc(vol.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT)
    desc = plain
    if with_map:
        smap = torch.zeros(capi.surface_map_bytes(plain) // 4, dtype=torch.int32, device="cuda")
        desc = capi.volume_desc(vol.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT, surface_map=smap.data_ptr())
        gpu_ctx.surface_map_rebuild(desc)
    _, R_v2c, t_v2c = synth.scene_s2(view)
    R, t = cam2vol(R_v2c, t_v2c)
    got, want = _run(gpu_ctx, desc, R, t, R_v2c, synth.INTR, synth.ROWS, synth.COLS, 0, 3)
    _check(got, want)
    # the maps hold surface and holes at every level, and level 1 is the oracle's resize of level 0
    for l in range(3):
        nan = np.isnan(want[l][0].view(np.float32)[..., 0])
        assert nan.any() and (~nan).sum() > 500
    p0, n0 = want[0][0].view(np.float32), want[0][1].view(np.float32)
    op, on = oracle.resize_points_normals(p0, n0)
    for o, g in ((op, got[1][0].view(np.float32)), (on, got[1][1].view(np.float32))):
        nan = np.isnan(o[..., 0])
        assert np.array_equal(np.isnan(g[..., 0]), nan) and np.array_equal(o[~nan].view(np.uint32), g[~nan].view(np.uint32))


@pytest.mark.parametrize("rows,cols,pad,levels", [(50, 70, 26, 3), (37, 53, 3, 3), (33, 31, 0, 2), (9, 14, 2, 3), (4, 4, 0, 3)])
def test_pyramid_odd_sizes_padded_rows_and_one_level(gpu_ctx, rows, cols, pad, levels):
    dims, vs, tr = synth.volume_geometry((96, 64, 80))
    vol_h = np.zeros(96 * 64 * 80, dtype=np.uint32)
    for f in (0, 40, 200):
        depth, R, t = synth.scene_s2(f)
        oracle.integrate(vol_h, dims, vs, tr, synth.MAX_WEIGHT, oracle.compute_dists(depth, synth.INTR), R, t, synth.INTR,
                         nthreads=8)
    vol = torch.from_numpy(vol_h.view(np.int32)).cuda()
    desc = capi.volume_desc(vol.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT)
    s = cols / 70.0
    intr = (64.0 * s, 61.0 * s, 34.2 * s, 25.1 * rows / 50.0)
    _, R_v2c, t_v2c = synth.scene_s2(20)
    R, t = cam2vol(R_v2c, t_v2c)
    got, want = _run(gpu_ctx, desc, R, t, R_v2c, intr, rows, cols, pad, levels)
    _check(got, want)
    assert (~np.isnan(want[0][0].view(np.float32)[..., 0])).any()


def test_pyramid_rejects_bad_arguments(gpu_ctx):
    vol = torch.zeros(16 ** 3, dtype=torch.int32, device="cuda")
    desc = capi.volume_desc(vol.data_ptr(), (16, 16, 16), (0.1, 0.1, 0.1), 0.3, 64)
    slab = capi.volume_desc(vol.data_ptr(), (16, 16, 32), (0.1, 0.1, 0.1), 0.3, 64, 0, 16)
    eye, zero, intr = np.eye(3).reshape(-1), np.zeros(3), (20.0, 20.0, 8.0, 8.0)
    p0, n0 = _maps(16, 16, 0)
    p1, n1 = _maps(8, 8, 0)
    p2, n2 = _maps(4, 4, 0)
    half = (p1.data_ptr(), 128, n1.data_ptr(), 128)
    call = lambda d, r, c, h, q: gpu_ctx.raycast_points_pyramid(d, eye, zero, eye, intr, p0.data_ptr(), 256, n0.data_ptr(), 256,  # noqa: E731
                                                                r, c, h, q)
    call(desc, 16, 16, half, (p2.data_ptr(), 64, n2.data_ptr(), 64))
    with pytest.raises(capi.YakHipError):
        call(desc, 16, 16, (None, 128, n1.data_ptr(), 128), None)             # no half-size points
    with pytest.raises(capi.YakHipError):
        call(desc, 16, 16, half, (p2.data_ptr(), 64, None, 64))               # quarter size: a pair or nothing
    with pytest.raises(capi.YakHipError):
        call(desc, 16, 16, (p1.data_ptr(), 120, n1.data_ptr(), 128), None)    # pitch below a row of 8 float4
    with pytest.raises(capi.YakHipError):
        call(desc, 1, 16, half, None)                                         # no half-size level of a one-row image
    with pytest.raises(capi.YakHipError):
        call(slab, 16, 16, half, None)                                        # whole volume only
    torch.cuda.synchronize()


@pytest.mark.parametrize("segments", [2, 4])
@pytest.mark.parametrize("rows,cols,pad", [(synth.ROWS, synth.COLS, 0), (50, 70, 26), (37, 53, 3), (9, 14, 2)])
def test_rays_marched_in_segments_give_the_same_maps(segments, rows, cols, pad):
    """Option raycast_segments (include/yakhip.h): every ray's pair sequence cut over 2 or 4 waves, the smallest pair
    index that ends a march wins -- tsdf_volume.cu:320-351 stops at the first such pair.  Full-size maps, both pyramid
    levels and the depth overload must carry the bits of the one-wave march (which the oracle pins elsewhere), image
    sizes that fill no tile included."""
    vol_h, dims, vs, tr = fused_volume(128, [0, 10, 20, 30, 40, 50, 120, 250], scene="s2")
    vol = torch.from_numpy(vol_h.view(np.int32)).cuda()
    plain = capi.volume_desc(vol.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT)
    smap = torch.zeros(capi.surface_map_bytes(plain) // 4, dtype=torch.int32, device="cuda")
    desc = capi.volume_desc(vol.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT, surface_map=smap.data_ptr())
    s = cols / synth.COLS
    intr = tuple(v * s for v in synth.INTR[:3]) + (synth.INTR[3] * rows / synth.ROWS,)
    ctx = capi.Context(0)
    try:
        ctx.surface_map_rebuild(desc)
        for view in (5, 133, 222):
            _, R_v2c, t_v2c = synth.scene_s2(view)
            R, t = cam2vol(R_v2c, t_v2c)
            pose = (np.asarray(R).reshape(-1), t, np.asarray(R_v2c).reshape(-1), intr)
            ctx.set_option("raycast_segments", 1)
            want, _ = _run(ctx, desc, R, t, R_v2c, intr, rows, cols, pad, 3)
            dn = _maps(rows, cols, pad)[0]
            dd = torch.full((rows, cols + pad), 7, dtype=torch.int16, device="cuda")
            ctx.raycast_depth(desc, *pose, dd.data_ptr(), (cols + pad) * 2, dn.data_ptr(), (cols + pad) * 16, rows, cols)
            ctx.set_option("raycast_segments", segments)
            got, _ = _run(ctx, desc, R, t, R_v2c, intr, rows, cols, pad, 3)
            dn2 = _maps(rows, cols, pad)[0]
            dd2 = torch.full((rows, cols + pad), 7, dtype=torch.int16, device="cuda")
            ctx.raycast_depth(desc, *pose, dd2.data_ptr(), (cols + pad) * 2, dn2.data_ptr(), (cols + pad) * 16, rows, cols)
            torch.cuda.synchronize()
            _check(got, want)
            assert torch.equal(dd, dd2) and torch.equal(dn.view(torch.int32), dn2.view(torch.int32))
            if rows == synth.ROWS:
                assert (~np.isnan(want[0][0].view(np.float32)[..., 0])).sum() > 10000
        with pytest.raises(capi.YakHipError):
            ctx.set_option("raycast_segments", 3)
    finally:
        ctx.close()
