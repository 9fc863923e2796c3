"""GPU parity: yakhip_raycast_points / yakhip_raycast_depth (through the C ABI) against the CPU oracle.

Bar (north_star): vertices within 1e-4 m of the oracle.  The kernel restates the same fp32 expression
tree (no contraction, correctly rounded '/' and sqrt), so it is in fact expected to be bit-identical;
the tests assert the stated tolerance AND bit equality, so a future divergence shows up as such.
The oracle's parity status for raycast is PARITY UNPINNED (the reference holds no vector for it).
"""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

import oracle  # noqa: E402
from yak_amd import capi, synth  # noqa: E402

pytestmark = pytest.mark.gpu

VERTEX_TOL_M = 1e-4   # north_star: "raycast vertices within 1e-4 m"
NORMAL_TOL = 1e-4


def cam2vol(R_v2c, t_v2c):
    R = np.ascontiguousarray(np.asarray(R_v2c, dtype=np.float32).T)
    t = (-(np.asarray(R_v2c, dtype=np.float64).T @ np.asarray(t_v2c, dtype=np.float64))).astype(np.float32)
    return R, t


def gpu_raycast(ctx, vol_host, dims, vs, tr, R, t, Rinv, intr, rows, cols, pitch_px=None, want_depth=False,
                step_factor=0.75, gradient_delta_factor=0.5, vol_dev=None):
    pitch = cols if pitch_px is None else pitch_px
    vol = torch.from_numpy(vol_host.view(np.int32)).cuda() if vol_dev is None else vol_dev
    desc = capi.volume_desc(vol.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT)
    nrm = torch.full((rows, pitch, 4), 7.0, dtype=torch.float32, device="cuda")
    if want_depth:
        out = torch.full((rows, pitch), 7, dtype=torch.int16, device="cuda")
        ctx.raycast_depth(desc, np.asarray(R).reshape(-1), t, np.asarray(Rinv).reshape(-1), intr, out.data_ptr(), pitch * 2,
                          nrm.data_ptr(), pitch * 16, rows, cols, step_factor, gradient_delta_factor)
    else:
        out = torch.full((rows, pitch, 4), 7.0, dtype=torch.float32, device="cuda")
        ctx.raycast_points(desc, np.asarray(R).reshape(-1), t, np.asarray(Rinv).reshape(-1), intr, out.data_ptr(), pitch * 16,
                           nrm.data_ptr(), pitch * 16, rows, cols, step_factor, gradient_delta_factor)
    torch.cuda.synchronize()
    out_h, nrm_h = out.cpu().numpy(), nrm.cpu().numpy()
    if pitch != cols:  # the padding columns must not be written
        assert (nrm_h[:, cols:] == 7.0).all() and (out_h[:, cols:] == 7).all()
    out_h = out_h[:, :cols].copy()
    if want_depth:
        out_h = out_h.view(np.uint16)
    return out_h, nrm_h[:, :cols].copy()


def assert_parity(got_p, got_n, want_p, want_n):
    nan_w = np.isnan(want_p[..., 0])
    assert np.array_equal(np.isnan(got_p[..., 0]), nan_w), "hit masks differ"
    assert np.array_equal(np.isnan(got_n[..., 0]), np.isnan(want_n[..., 0]))
    assert np.isnan(got_p[nan_w]).all() and np.isnan(got_n[nan_w]).all()
    if (~nan_w).any():
        assert np.abs(got_p[~nan_w] - want_p[~nan_w]).max() <= VERTEX_TOL_M
        assert np.abs(got_n[~nan_w] - want_n[~nan_w]).max() <= NORMAL_TOL
    # stronger than the stated bar: same bits
    assert np.array_equal(got_p[~nan_w].view(np.uint32), want_p[~nan_w].view(np.uint32))
    assert np.array_equal(got_n[~nan_w].view(np.uint32), want_n[~nan_w].view(np.uint32))


def fused_volume(n, frames, scene="s1"):
    dims, vs, tr = synth.volume_geometry(n)
    vol = np.zeros(int(np.prod(dims)), dtype=np.uint32)
    for f in frames:
        depth, R, t = synth.scene_s1() if scene == "s1" else synth.scene_s2(f)
        dists = oracle.compute_dists(depth, synth.INTR)
        oracle.integrate(vol, dims, vs, tr, synth.MAX_WEIGHT, dists, R, t, synth.INTR, nthreads=8)
    return vol, dims, vs, tr


@pytest.mark.parametrize("n", [32, 64, 128])
def test_raycast_s1_from_fusing_pose(gpu_ctx, n):
    vol, dims, vs, tr = fused_volume(n, [0, 0])
    _, R_v2c, t_v2c = synth.scene_s1()
    R, t = cam2vol(R_v2c, t_v2c)
    want_p, want_n, _, hits = oracle.raycast(vol, dims, vs, tr, R, t, R_v2c, synth.INTR, synth.ROWS, synth.COLS, nthreads=8)
    assert hits > 1000
    got_p, got_n = gpu_raycast(gpu_ctx, vol, dims, vs, tr, R, t, R_v2c, synth.INTR, synth.ROWS, synth.COLS)
    assert_parity(got_p, got_n, want_p, want_n)


@pytest.mark.parametrize("view", [5, 47, 133, 260])
def test_raycast_s2_orbit_views(gpu_ctx, view):
    vol, dims, vs, tr = fused_volume(128, [0, 10, 20, 30, 40, 50, 120, 250], scene="s2")
    _, R_v2c, t_v2c = synth.scene_s2(view)
    R, t = cam2vol(R_v2c, t_v2c)
    want_p, want_n, _, hits = oracle.raycast(vol, dims, vs, tr, R, t, R_v2c, synth.INTR, synth.ROWS, synth.COLS, nthreads=8)
    assert hits > 1000
    got_p, got_n = gpu_raycast(gpu_ctx, vol, dims, vs, tr, R, t, R_v2c, synth.INTR, synth.ROWS, synth.COLS)
    assert_parity(got_p, got_n, want_p, want_n)


def test_raycast_depth_overload(gpu_ctx):
    vol, dims, vs, tr = fused_volume(64, [0, 0])
    _, R_v2c, t_v2c = synth.scene_s1()
    R, t = cam2vol(R_v2c, t_v2c)
    want_d, want_n, _, hits = oracle.raycast(vol, dims, vs, tr, R, t, R_v2c, synth.INTR, synth.ROWS, synth.COLS,
                                             want_depth=True, nthreads=8)
    got_d, got_n = gpu_raycast(gpu_ctx, vol, dims, vs, tr, R, t, R_v2c, synth.INTR, synth.ROWS, synth.COLS, want_depth=True)
    assert hits > 1000 and np.array_equal(got_d, want_d)
    nan = np.isnan(want_n[..., 0])
    assert np.array_equal(np.isnan(got_n[..., 0]), nan)
    assert np.array_equal(got_n[~nan].view(np.uint32), want_n[~nan].view(np.uint32))


def test_raycast_ragged_image_pitched_rows_noncubic_volume(gpu_ctx):
    # image 70 x 50 (not a multiple of the 16 x 16 tile), padded rows, volume 96 x 64 x 80
    dims, vs, tr = synth.volume_geometry((96, 64, 80))
    vol = np.zeros(96 * 64 * 80, dtype=np.uint32)
    for f in (0, 40, 200):
        depth, R, t = synth.scene_s2(f)
        oracle.integrate(vol, dims, vs, tr, synth.MAX_WEIGHT, oracle.compute_dists(depth, synth.INTR), R, t, synth.INTR,
                         nthreads=8)
    rows, cols = 50, 70
    intr = (64.0, 61.0, 34.2, 25.1)
    _, R_v2c, t_v2c = synth.scene_s2(20)
    R, t = cam2vol(R_v2c, t_v2c)
    want_p, want_n, _, hits = oracle.raycast(vol, dims, vs, tr, R, t, R_v2c, intr, rows, cols)
    assert hits > 100
    got_p, got_n = gpu_raycast(gpu_ctx, vol, dims, vs, tr, R, t, R_v2c, intr, rows, cols, pitch_px=96)
    assert_parity(got_p, got_n, want_p, want_n)
    # other step / gradient factors
    want_p, want_n, _, _ = oracle.raycast(vol, dims, vs, tr, R, t, R_v2c, intr, rows, cols, step_factor=0.4,
                                          gradient_delta_factor=1.25)
    got_p, got_n = gpu_raycast(gpu_ctx, vol, dims, vs, tr, R, t, R_v2c, intr, rows, cols, step_factor=0.4,
                               gradient_delta_factor=1.25)
    assert_parity(got_p, got_n, want_p, want_n)


def test_raycast_camera_inside_and_misses(gpu_ctx):
    vol, dims, vs, tr = fused_volume(64, [0, 60, 150, 240], scene="s2")
    # camera inside the volume (tmin clamps to 0, cu:311-312)
    R_c2v, eye = synth.look_at((1.5, 0.6, 0.3), (1.3, 1.6, 1.4))
    R_v2c, t_v2c = synth.vol2cam(R_c2v, eye)
    R, t = cam2vol(R_v2c, t_v2c)
    want_p, want_n, _, hits = oracle.raycast(vol, dims, vs, tr, R, t, R_v2c, synth.INTR, synth.ROWS, synth.COLS, nthreads=8)
    assert hits > 1000
    got_p, got_n = gpu_raycast(gpu_ctx, vol, dims, vs, tr, R, t, R_v2c, synth.INTR, synth.ROWS, synth.COLS)
    assert_parity(got_p, got_n, want_p, want_n)
    # looking away: no ray enters the box
    R_c2v, eye = synth.look_at((1.5, 1.5, -2.0), (1.5, 1.5, -5.0))
    R_v2c, t_v2c = synth.vol2cam(R_c2v, eye)
    R, t = cam2vol(R_v2c, t_v2c)
    got_p, got_n = gpu_raycast(gpu_ctx, vol, dims, vs, tr, R, t, R_v2c, synth.INTR, synth.ROWS, synth.COLS)
    assert np.isnan(got_p).all() and np.isnan(got_n).all()
    # cleared volume: rays march through, nothing is found
    empty = np.zeros_like(vol)
    _, R_v2c, t_v2c = synth.scene_s1()
    R, t = cam2vol(R_v2c, t_v2c)
    got_p, got_n = gpu_raycast(gpu_ctx, empty, dims, vs, tr, R, t, R_v2c, synth.INTR, synth.ROWS, synth.COLS)
    assert np.isnan(got_p).all() and np.isnan(got_n).all()
    got_d, _ = gpu_raycast(gpu_ctx, empty, dims, vs, tr, R, t, R_v2c, synth.INTR, synth.ROWS, synth.COLS, want_depth=True)
    assert not got_d.any()


def test_raycast_golden_fixture(gpu_ctx):
    """Oracle-free vector: tests/golden/raycast_s1_32.npz (see tests/golden/make_golden.py)."""
    here = os.path.dirname(__file__)
    g = np.load(os.path.join(here, "golden", "raycast_s1_32.npz"))
    v = np.load(os.path.join(here, "golden", "s1_32.npz"))
    dims, vs, tr = synth.volume_geometry(32)
    got_p, got_n = gpu_raycast(gpu_ctx, np.ascontiguousarray(v["vol_f3"]), dims, vs, tr, g["R"], g["t"], g["Rinv"],
                               g["intr"], 60, 80)
    assert_parity(got_p, got_n, g["points"].view(np.float32), g["normals"].view(np.float32))


def test_raycast_512_volume_fused_on_gpu(gpu_ctx):
    """BASELINE config size: 512^3, S1 fused twice by the HIP integrate, ray-cast by both sides."""
    n = 512
    dims, vs, tr = synth.volume_geometry(n)
    depth, R_v2c, t_v2c = synth.scene_s1()
    d_depth = torch.from_numpy(depth.view(np.int16)).cuda()
    d_dists = torch.empty_like(d_depth)
    vol = torch.zeros(n ** 3, dtype=torch.int32, device="cuda")
    desc = capi.volume_desc(vol.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT)
    gpu_ctx.compute_dists(d_depth.data_ptr(), synth.COLS * 2, d_dists.data_ptr(), synth.COLS * 2, synth.ROWS, synth.COLS,
                          synth.INTR)
    for _ in range(2):
        gpu_ctx.integrate(desc, d_dists.data_ptr(), synth.COLS * 2, synth.ROWS, synth.COLS, R_v2c.reshape(-1), t_v2c,
                          synth.INTR)
    torch.cuda.synchronize()
    vol_h = vol.cpu().numpy().view(np.uint32)
    for view in (None, 30):
        if view is not None:
            _, R_v2c, t_v2c = synth.scene_s2(view)
        R, t = cam2vol(R_v2c, t_v2c)
        want_p, want_n, _, hits = oracle.raycast(vol_h, dims, vs, tr, R, t, R_v2c, synth.INTR, synth.ROWS, synth.COLS,
                                                 nthreads=16)
        assert hits > 10000
        got_p, got_n = gpu_raycast(gpu_ctx, None, dims, vs, tr, R, t, R_v2c, synth.INTR, synth.ROWS, synth.COLS, vol_dev=vol)
        assert_parity(got_p, got_n, want_p, want_n)
        # physical check, independent of the oracle: central sphere pixels within two voxels of the
        # frame (one for the grid, one for the half-precision ray lengths: 3.9 mm spacing at 4 m)
        if view is None:
            u, v = np.meshgrid(np.arange(synth.COLS), np.arange(synth.ROWS))
            inner = np.hypot(u - synth.INTR[2], v - synth.INTR[3]) < 60
            assert not np.isnan(got_p[inner]).any()
            assert np.abs(got_p[inner][:, 2] - depth[inner] / 1000.0).max() < 2 * vs[0]


def test_raycast_rejects_bad_arguments(gpu_ctx):
    dims, vs, tr = synth.volume_geometry(16)
    vol = torch.zeros(16 ** 3, dtype=torch.int32, device="cuda")
    img = torch.zeros((8, 8, 4), dtype=torch.float32, device="cuda")
    eye = np.eye(3, dtype=np.float32).reshape(-1)
    t = np.zeros(3, dtype=np.float32)
    slab = capi.volume_desc(vol.data_ptr(), dims, vs, tr, 64, 0, 8)
    with pytest.raises(capi.YakHipError) as e:
        gpu_ctx.raycast_points(slab, eye, t, eye, synth.INTR, img.data_ptr(), 128, img.data_ptr(), 128, 8, 8)
    assert e.value.code == -1 and "whole volume" in str(e.value)
    whole = capi.volume_desc(vol.data_ptr(), dims, vs, tr, 64)
    with pytest.raises(capi.YakHipError) as e:   # pitch smaller than a row
        gpu_ctx.raycast_points(whole, eye, t, eye, synth.INTR, img.data_ptr(), 64, img.data_ptr(), 128, 8, 8)
    assert e.value.code == -1
    with pytest.raises(capi.YakHipError) as e:   # misaligned float4 rows
        gpu_ctx.raycast_points(whole, eye, t, eye, synth.INTR, img.data_ptr() + 4, 128, img.data_ptr(), 128, 8, 8)
    assert e.value.code == -2
    with pytest.raises(capi.YakHipError) as e:   # null output
        gpu_ctx.raycast_points(whole, eye, t, eye, synth.INTR, 0, 128, img.data_ptr(), 128, 8, 8)
    assert e.value.code == -1
    with pytest.raises(capi.YakHipError) as e:
        gpu_ctx.raycast_points(whole, eye, t, eye, synth.INTR, img.data_ptr(), 128, img.data_ptr(), 128, 8, 8,
                               step_factor=0.0)
    assert e.value.code == -1


def _looking_along_z(eye):
    """cam2vol of a camera at `eye` whose axes are the volume's: pixel (cy, cx) looks exactly along +z."""
    return np.eye(3, dtype=np.float32), np.asarray(eye, dtype=np.float32)


RAY_CASES = {
    # rays with a zero direction component (1 / 0 = inf in the slab test, cu:149-165) ...
    "axis-aligned central ray": lambda R, t: _looking_along_z((1.47, 1.53, -1.0)),
    # ... and the camera exactly on a face plane of the box besides: inf * 0 = NaN in the slab test
    "axis-aligned ray in the x = 0 plane": lambda R, t: _looking_along_z((0.0, 1.5, -1.0)),
    "camera on the z = 0 face": lambda R, t: _looking_along_z((1.5, 1.5, 0.0)),
    "camera on the far corner of the box": lambda R, t: (R, np.array([3.0, 3.0, 3.0], dtype=np.float32) * np.float32(63.0 / 64.0)),
    "scaled rotation": lambda R, t: (np.float32(2.5) * R, t),
    "mirrored": lambda R, t: (R * np.array([[1.0, -1.0, 1.0]], dtype=np.float32), t),
    "zero rotation": lambda R, t: (np.zeros((3, 3), dtype=np.float32), t),
    "NaN translation": lambda R, t: (R, np.array([t[0], np.nan, t[2]], dtype=np.float32)),
    "NaN rotation entry": lambda R, t: (np.where(np.eye(3) > 0, R, np.float32(np.nan)).astype(np.float32), t),
    "infinite translation": lambda R, t: (R, np.array([np.inf, t[1], t[2]], dtype=np.float32)),
    "camera a thousand kilometres away": lambda R, t: (R, (t - np.float32(1e6) * R[:, 2]).astype(np.float32)),
}


@pytest.mark.parametrize("with_map", [False, True])
@pytest.mark.parametrize("case", sorted(RAY_CASES))
def test_raycast_degenerate_rays_and_poses(gpu_ctx, case, with_map):
    """Rays the slab test divides by zero for, cameras on the box, poses no camera has: the maps must be the oracle's
    (tsdf_volume.cu:149-165, 288-352 with IEEE inf / NaN arithmetic), with the surface map and without."""
    vol_h, dims, vs, tr = fused_volume(64, [0, 60, 150, 240], scene="s2")
    _, R_v2c, t_v2c = synth.scene_s2(20)
    R0, t0 = cam2vol(R_v2c, t_v2c)
    R, t = RAY_CASES[case](np.asarray(R0, dtype=np.float32).reshape(3, 3), np.asarray(t0, dtype=np.float32))
    Rinv = np.ascontiguousarray(np.asarray(R_v2c, dtype=np.float32))
    rows, cols, intr = 120, 160, (131.0, 131.0, 80.0, 60.0)  # integer principal point: pixel (60, 80) has rx = ry = 0
    want_p, want_n, _, hits = oracle.raycast(vol_h, dims, vs, tr, R, t, Rinv, intr, rows, cols, nthreads=8)
    vol = torch.from_numpy(vol_h.view(np.int32)).cuda()
    if with_map:
        plain = capi.volume_desc(vol.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT)
        smap = torch.zeros(capi.surface_map_bytes(plain) // 4, dtype=torch.int32, device="cuda")
        desc = capi.volume_desc(vol.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT, surface_map=smap.data_ptr())
        gpu_ctx.surface_map_rebuild(desc)
    else:
        desc = capi.volume_desc(vol.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT)
    pts = torch.full((rows, cols, 4), 7.0, dtype=torch.float32, device="cuda")
    nrm = torch.full((rows, cols, 4), 7.0, dtype=torch.float32, device="cuda")
    gpu_ctx.raycast_points(desc, R.reshape(-1), t, Rinv.reshape(-1), intr, pts.data_ptr(), cols * 16, nrm.data_ptr(), cols * 16,
                           rows, cols)
    torch.cuda.synchronize()
    assert_parity(pts.cpu().numpy(), nrm.cpu().numpy(), want_p, want_n)
    if case in ("axis-aligned central ray", "scaled rotation", "mirrored", "camera on the z = 0 face"):
        assert hits > 100, "%s: the case is meant to see the surface (%d hits)" % (case, hits)


@pytest.mark.parametrize("with_map", [False, True])
@pytest.mark.parametrize("finite", [True, False])
def test_raycast_through_a_volume_of_arbitrary_words(gpu_ctx, finite, with_map):
    """Any voxel words (uploadTSDF): every sample pair is a candidate crossing, interpolation runs on huge and
    denormal halves and, with finite=False, on infinities and NaNs.  Same maps as the oracle, lane for lane: NaN where
    it has NaN, the same bits elsewhere."""
    from test_mc_gpu import arbitrary_words

    dims, vs, tr = synth.volume_geometry((48, 40, 56))
    vol_h = arbitrary_words(dims, 99, finite)
    _, R_v2c, t_v2c = synth.scene_s2(20)
    R, t = cam2vol(R_v2c, t_v2c)
    rows, cols, intr = 120, 160, (131.0, 131.0, 79.5, 59.5)
    want_p, want_n, _, hits = oracle.raycast(vol_h, dims, vs, tr, R, t, R_v2c, intr, rows, cols, nthreads=8)
    assert hits > 1000
    vol = torch.from_numpy(vol_h.view(np.int32)).cuda()
    desc = capi.volume_desc(vol.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT)
    if with_map:
        smap = torch.zeros(capi.surface_map_bytes(desc) // 4, dtype=torch.int32, device="cuda")
        desc = capi.volume_desc(vol.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT, surface_map=smap.data_ptr())
        gpu_ctx.surface_map_rebuild(desc)
    pts = torch.full((rows, cols, 4), 7.0, dtype=torch.float32, device="cuda")
    nrm = torch.full((rows, cols, 4), 7.0, dtype=torch.float32, device="cuda")
    gpu_ctx.raycast_points(desc, np.asarray(R).reshape(-1), t, np.asarray(R_v2c).reshape(-1), intr, pts.data_ptr(), cols * 16,
                           nrm.data_ptr(), cols * 16, rows, cols)
    torch.cuda.synchronize()
    for got, want in ((pts.cpu().numpy(), want_p), (nrm.cpu().numpy(), want_n)):
        nan = np.isnan(want)
        assert np.array_equal(np.isnan(got), nan)
        assert np.array_equal(got[~nan].view(np.uint32), want[~nan].view(np.uint32))
