"""GPU parity tests for the hot path: HIP kernels (through the C ABI) vs the scalar CPU oracle.

Bar (BASELINE.json north_star): voxel weights bit-exact, tsdf within 1 half-precision ulp.
The kernels use the same IEEE operations in the same order as the oracle, so these tests assert the
stricter thing first -- identical 32-bit voxel words -- and state the north-star tolerance beside it.
"""
import os

import numpy as np
import pytest

import oracle
from yak_amd import capi, synth

from gpu_helpers import gpu_compute_dists, gpu_integrate, half_ulp_distance

pytestmark = pytest.mark.gpu

TSDF_TOL_HALF_ULP = 1  # north_star: "tsdf values must match within 1 half-precision ulp"


def assert_parity(got, want):
    assert got.shape == want.shape
    assert np.array_equal(got >> 16, want >> 16), "voxel weights differ (must be bit-exact)"
    ulp = half_ulp_distance((got & 0xFFFF).astype(np.uint16), (want & 0xFFFF).astype(np.uint16))
    assert ulp.max() <= TSDF_TOL_HALF_ULP, "tsdf differs by %d half ulps" % ulp.max()
    assert np.array_equal(got, want), "%d voxel words differ (within tolerance but not bit-exact)" % np.count_nonzero(
        got != want)


def scene(name, frame=0):
    if name == "s0":
        return synth.scene_s0()
    if name == "s1":
        return synth.scene_s1()
    return synth.scene_s2(frame)


# ---------------------------------------------------------------------------- compute_dists
@pytest.mark.parametrize("name,frame", [("s0", 0), ("s1", 0), ("s2", 0), ("s2", 133)])
def test_compute_dists_bit_exact(gpu_ctx, name, frame):
    depth, _, _ = scene(name, frame)
    want = oracle.compute_dists(depth, synth.INTR)
    assert np.array_equal(gpu_compute_dists(gpu_ctx, depth, synth.INTR), want)


def test_compute_dists_random_depth_pitched_and_odd_width(gpu_ctx):
    rng = np.random.default_rng(3)
    for rows, cols, pitch_in, pitch_out in [(480, 640, 704, 768), (37, 53, 53, 55), (5, 7, 9, 7), (1, 1, 1, 1)]:
        depth = rng.integers(0, 65536, size=(rows, cols), dtype=np.uint16)
        depth[rng.random((rows, cols)) < 0.2] = 0
        intr = (525.0, 530.0, cols / 2 - 0.25, rows / 2 + 0.125)
        want = oracle.compute_dists(depth, intr)
        got = gpu_compute_dists(gpu_ctx, depth, intr, pitch_in, pitch_out)
        assert np.array_equal(got, want), (rows, cols)


# ---------------------------------------------------------------------------- integrate
@pytest.mark.parametrize("name,frame,n", [("s0", 0, 64), ("s1", 0, 64), ("s1", 0, 128), ("s2", 0, 96),
                                          ("s2", 77, 128), ("s2", 250, 64)])
def test_integrate_single_frame_bit_exact(gpu_ctx, name, frame, n):
    depth, R, t = scene(name, frame)
    dists = oracle.compute_dists(depth, synth.INTR)
    dims, vs, tr = synth.volume_geometry(n)
    want = np.zeros(n ** 3, dtype=np.uint32)
    _, updated = oracle.integrate(want, dims, vs, tr, synth.MAX_WEIGHT, dists, R, t, synth.INTR,
                                  nthreads=oracle.max_threads())
    assert updated > 0
    got = gpu_integrate(gpu_ctx, np.zeros(n ** 3, dtype=np.uint32), dims, vs, tr, synth.MAX_WEIGHT, dists, R, t,
                        synth.INTR)
    assert_parity(got, want)


def test_integrate_orbit_sequence_accumulates_bit_exact(gpu_ctx):
    """12 frames of the S2 orbit fused one after another: running averages, mixed weights."""
    n = 80
    dims, vs, tr = synth.volume_geometry(n)
    want = np.zeros(n ** 3, dtype=np.uint32)
    got = np.zeros(n ** 3, dtype=np.uint32)
    for frame in range(0, 120, 10):
        depth, R, t = synth.scene_s2(frame, noise_mm=1.0)
        dists = oracle.compute_dists(depth, synth.INTR)
        oracle.integrate(want, dims, vs, tr, synth.MAX_WEIGHT, dists, R, t, synth.INTR, nthreads=oracle.max_threads())
        got = gpu_integrate(gpu_ctx, got, dims, vs, tr, synth.MAX_WEIGHT, dists, R, t, synth.INTR)
        assert_parity(got, want)
    assert len(np.unique(want >> 16)) > 4


def test_integrate_weight_saturation(gpu_ctx):
    depth, R, t = synth.scene_s1()
    dists = oracle.compute_dists(depth, synth.INTR)
    dims, vs, tr = synth.volume_geometry(32)
    want = np.zeros(32 ** 3, dtype=np.uint32)
    for _ in range(6):
        oracle.integrate(want, dims, vs, tr, 4, dists, R, t, synth.INTR)
    got = gpu_integrate(gpu_ctx, np.zeros(32 ** 3, dtype=np.uint32), dims, vs, tr, 4, dists, R, t, synth.INTR,
                        frames=6)
    assert (want >> 16).max() == 4
    assert_parity(got, want)


@pytest.mark.parametrize("dims", [(33, 17, 29), (30, 22, 19), (64, 48, 40), (1, 1, 10), (4, 1, 3), (2, 3, 5)])
def test_integrate_ragged_dims_all_vector_widths(gpu_ctx, dims):
    """X % 4 == 0 takes the 16-byte path, X % 2 == 0 the 8-byte path, odd X the scalar path."""
    depth, R, t = synth.scene_s2(40)
    dists = oracle.compute_dists(depth, synth.INTR)
    _, vs, tr = synth.volume_geometry(dims)
    n = dims[0] * dims[1] * dims[2]
    want = np.zeros(n, dtype=np.uint32)
    oracle.integrate(want, dims, vs, tr, synth.MAX_WEIGHT, dists, R, t, synth.INTR)
    got = gpu_integrate(gpu_ctx, np.zeros(n, dtype=np.uint32), dims, vs, tr, synth.MAX_WEIGHT, dists, R, t, synth.INTR)
    assert_parity(got, want)


@pytest.mark.parametrize("misalign", [1, 2, 3])
def test_integrate_unaligned_base_pointer(gpu_ctx, misalign):
    depth, R, t = synth.scene_s1()
    dists = oracle.compute_dists(depth, synth.INTR)
    dims, vs, tr = synth.volume_geometry(32)
    want = np.zeros(32 ** 3, dtype=np.uint32)
    oracle.integrate(want, dims, vs, tr, synth.MAX_WEIGHT, dists, R, t, synth.INTR)
    got = gpu_integrate(gpu_ctx, np.zeros(32 ** 3, dtype=np.uint32), dims, vs, tr, synth.MAX_WEIGHT, dists, R, t,
                        synth.INTR, misalign_words=misalign)
    assert_parity(got, want)


def test_integrate_pitched_dists_image(gpu_ctx):
    depth, R, t = synth.scene_s2(5)
    dists = oracle.compute_dists(depth, synth.INTR)
    dims, vs, tr = synth.volume_geometry(48)
    want = np.zeros(48 ** 3, dtype=np.uint32)
    oracle.integrate(want, dims, vs, tr, synth.MAX_WEIGHT, dists, R, t, synth.INTR)
    got = gpu_integrate(gpu_ctx, np.zeros(48 ** 3, dtype=np.uint32), dims, vs, tr, synth.MAX_WEIGHT, dists, R, t,
                        synth.INTR, dists_pitch_elems=768)
    assert_parity(got, want)


@pytest.mark.parametrize("z_chunk", [1, 2, 7, 16, 1000])
def test_integrate_any_z_chunk_gives_the_same_volume(gpu_ctx, z_chunk):
    """The z decomposition replays the reference's float accumulation, so it must not change a bit."""
    depth, R, t = synth.scene_s2(60)
    dists = oracle.compute_dists(depth, synth.INTR)
    dims, vs, tr = synth.volume_geometry((40, 36, 50))
    n = 40 * 36 * 50
    want = np.zeros(n, dtype=np.uint32)
    oracle.integrate(want, dims, vs, tr, synth.MAX_WEIGHT, dists, R, t, synth.INTR)
    gpu_ctx.set_option("z_chunk", z_chunk)
    try:
        got = gpu_integrate(gpu_ctx, np.zeros(n, dtype=np.uint32), dims, vs, tr, synth.MAX_WEIGHT, dists, R, t,
                            synth.INTR)
    finally:
        gpu_ctx.set_option("z_chunk", 0)
    assert_parity(got, want)


def test_integrate_z_slabs_concatenate_to_the_whole_volume(gpu_ctx):
    """Multi-GPU sharding contract: each rank integrates layers [z0, z1); the concatenation is
    bit-identical to a single whole-volume integrate (no halo, no exchange)."""
    depth, R, t = synth.scene_s2(200)
    dists = oracle.compute_dists(depth, synth.INTR)
    dims, vs, tr = synth.volume_geometry(64)
    plane = 64 * 64
    want = np.zeros(64 ** 3, dtype=np.uint32)
    oracle.integrate(want, dims, vs, tr, synth.MAX_WEIGHT, dists, R, t, synth.INTR, nthreads=oracle.max_threads())
    parts = []
    for z0, z1 in [(0, 8), (8, 16), (16, 37), (37, 64)]:
        slab = gpu_integrate(gpu_ctx, np.zeros(plane * (z1 - z0), dtype=np.uint32), dims, vs, tr, synth.MAX_WEIGHT,
                             dists, R, t, synth.INTR, z0=z0, z1=z1)
        parts.append(slab)
    assert_parity(np.concatenate(parts), want)


def test_integrate_nontemporal_switch_does_not_change_results(gpu_ctx):
    depth, R, t = synth.scene_s1()
    dists = oracle.compute_dists(depth, synth.INTR)
    dims, vs, tr = synth.volume_geometry(48)
    a = gpu_integrate(gpu_ctx, np.zeros(48 ** 3, dtype=np.uint32), dims, vs, tr, 64, dists, R, t, synth.INTR)
    gpu_ctx.set_option("nontemporal", 0)
    try:
        b = gpu_integrate(gpu_ctx, np.zeros(48 ** 3, dtype=np.uint32), dims, vs, tr, 64, dists, R, t, synth.INTR)
    finally:
        gpu_ctx.set_option("nontemporal", 1)
    assert np.array_equal(a, b)


def test_integrate_empty_frame_and_camera_behind(gpu_ctx):
    dims, vs, tr = synth.volume_geometry(32)
    zero = np.zeros(32 ** 3, dtype=np.uint32)
    _, R, t = synth.scene_s0()
    # no returns at all: Dp == 0 everywhere
    got = gpu_integrate(gpu_ctx, zero, dims, vs, tr, 64, np.zeros((480, 640), dtype=np.uint16), R, t, synth.INTR)
    assert not got.any()
    # camera looking away from the volume: vc.z <= 0 everywhere
    dists = oracle.compute_dists(synth.scene_s0()[0], synth.INTR)
    got = gpu_integrate(gpu_ctx, zero, dims, vs, tr, 64, dists, R, np.array([-1.5, -1.5, -9.0], dtype=np.float32),
                        synth.INTR)
    assert not got.any()


def test_golden_fixture_without_the_oracle(gpu_ctx):
    """Committed vectors (tests/golden/s1_32.npz): the HIP path reproduces them word for word."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "s1_32.npz"))
    depth, R, t = synth.scene_s1()
    dists = gpu_compute_dists(gpu_ctx, depth, synth.INTR)
    dims, vs, tr = synth.volume_geometry(32)
    f1 = gpu_integrate(gpu_ctx, np.zeros(32 ** 3, dtype=np.uint32), dims, vs, tr, 64, dists, R, t, synth.INTR)
    assert np.array_equal(f1, g["vol_f1"])
    f3 = gpu_integrate(gpu_ctx, f1, dims, vs, tr, 64, dists, R, t, synth.INTR, frames=2)
    assert np.array_equal(f3, g["vol_f3"])


def test_clear_volume(gpu_ctx):
    import torch

    for n, off in [(64 ** 3, 0), (1000, 1), (3, 3), (17, 2)]:
        store = torch.full((n + 8,), 0x7F7F7F7F, dtype=torch.int32, device="cuda")
        desc = capi.volume_desc(store[off:].data_ptr(), (n, 1, 1), (0.1, 0.1, 0.1), 0.3, 64)
        gpu_ctx.clear_volume(desc)
        torch.cuda.synchronize()
        host = store.cpu().numpy()
        assert not host[off:off + n].any()
        assert np.all(host[:off] == 0x7F7F7F7F) and np.all(host[off + n:] == 0x7F7F7F7F)


def test_rejected_arguments_raise(gpu_ctx):
    import torch

    buf = torch.zeros(64, dtype=torch.int32, device="cuda")
    img = torch.zeros((4, 4), dtype=torch.int16, device="cuda")
    R, t = np.eye(3).reshape(-1), (0, 0, 0)
    ok = capi.volume_desc(buf.data_ptr(), (4, 4, 4), (0.1, 0.1, 0.1), 0.3, 64)
    gpu_ctx.integrate(ok, img.data_ptr(), 8, 4, 4, R, t, synth.INTR)
    for bad in [capi.volume_desc(0, (4, 4, 4), (0.1, 0.1, 0.1), 0.3, 64),
                capi.volume_desc(buf.data_ptr(), (0, 4, 4), (0.1, 0.1, 0.1), 0.3, 64),
                capi.volume_desc(buf.data_ptr(), (4, 4, 4), (0.1, 0.1, 0.1), 0.3, 64, z0=3, z1=2),
                capi.volume_desc(buf.data_ptr(), (4, 4, 4), (0.1, 0.1, 0.1), 0.3, 64, z0=0, z1=5),
                capi.volume_desc(buf.data_ptr(), (4, 4, 4), (0.1, 0.1, 0.1), 0.3, 40000),
                capi.volume_desc(buf.data_ptr(), (4, 4, 4), (0.1, 0.1, 0.1), 0.0, 64)]:
        with pytest.raises(capi.YakHipError):
            gpu_ctx.integrate(bad, img.data_ptr(), 8, 4, 4, R, t, synth.INTR)
    with pytest.raises(capi.YakHipError):
        gpu_ctx.integrate(ok, img.data_ptr(), 6, 4, 4, R, t, synth.INTR)  # pitch smaller than a row
    with pytest.raises(capi.YakHipError):
        gpu_ctx.integrate(ok, 0, 8, 4, 4, R, t, synth.INTR)
    torch.cuda.synchronize()


# ---------------------------------------------------------------------------- full size (512^3)
def test_full_size_s0_known_answer_and_idempotent_weights(gpu_ctx):
    """BASELINE config 2 at full size, checked by a size-independent property: in scene S0 every
    voxel is in view and in front of the wall by more than trunc, so after k frames every voxel is
    exactly (tsdf 1.0, weight k)."""
    import torch

    n = 512
    dims, vs, tr = synth.volume_geometry(n)
    depth, R, t = synth.scene_s0()
    dists = gpu_compute_dists(gpu_ctx, depth, synth.INTR)
    dd = torch.from_numpy(dists.view(np.int16)).cuda()
    vol = torch.empty(n ** 3, dtype=torch.int32, device="cuda")
    desc = capi.volume_desc(vol.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT)
    gpu_ctx.clear_volume(desc)
    for k in range(1, 4):
        gpu_ctx.integrate(desc, dd.data_ptr(), 640 * 2, 480, 640, R.reshape(-1), t, synth.INTR)
        torch.cuda.synchronize()
        assert bool((vol == ((k << 16) | 0x3C00)).all()), "frame %d" % k


def test_full_size_s1_against_threaded_oracle(gpu_ctx):
    """512^3, scene S1, one frame: the whole 512 MiB volume compared word for word with the oracle
    (OpenMP over the box's host cores; a few seconds)."""
    import torch

    n = 512
    dims, vs, tr = synth.volume_geometry(n)
    depth, R, t = synth.scene_s1()
    dists = oracle.compute_dists(depth, synth.INTR)
    want = np.zeros(n ** 3, dtype=np.uint32)
    oracle.integrate(want, dims, vs, tr, synth.MAX_WEIGHT, dists, R, t, synth.INTR,
                     nthreads=min(16, oracle.max_threads()))
    dd = torch.from_numpy(dists.view(np.int16)).cuda()
    vol = torch.zeros(n ** 3, dtype=torch.int32, device="cuda")
    desc = capi.volume_desc(vol.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT)
    gpu_ctx.integrate(desc, dd.data_ptr(), 640 * 2, 480, 640, R.reshape(-1), t, synth.INTR)
    torch.cuda.synchronize()
    got = vol.cpu().numpy().view(np.uint32)
    assert_parity(got, want)


# ---------------------------------------------------------------------------- randomized poses
def _random_pose(rng, inside):
    """Camera somewhere around (or inside) the 3 m cube, looking roughly at it, with roll."""
    centre = np.array([1.5, 1.5, 1.5])
    if inside:
        eye = centre + rng.uniform(-1.2, 1.2, 3)
        target = eye + rng.normal(size=3)
    else:
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        eye = centre + d * rng.uniform(2.2, 6.0)
        target = centre + rng.uniform(-0.8, 0.8, 3)
    up = rng.normal(size=3)
    R_c2v, eye = synth.look_at(eye, target, up=up / np.linalg.norm(up))
    return R_c2v, eye


@pytest.mark.parametrize("seed", range(12))
def test_integrate_random_poses_scenes_and_intrinsics(gpu_ctx, seed):
    """Cameras outside, inside and grazing the volume, rolled, with odd intrinsics and image sizes:
    exercises the group test's margins (image borders, depth edges, near plane), the band path and the
    exact fallbacks against the oracle, three frames deep."""
    rng = np.random.default_rng(1000 + seed)
    rows, cols = [(480, 640), (240, 320), (97, 131), (600, 800)][seed % 4]
    f = cols * rng.uniform(0.6, 1.6)
    intr = (float(f), float(f * rng.uniform(0.9, 1.1)), float(cols / 2 + rng.uniform(-20, 20)),
            float(rows / 2 + rng.uniform(-20, 20)))
    n = [64, 48, 80, 56][seed % 4]
    dims, vs, tr = synth.volume_geometry((n, n - 8, n + 4))
    nvox = dims[0] * dims[1] * dims[2]
    want = np.zeros(nvox, dtype=np.uint32)
    got = np.zeros(nvox, dtype=np.uint32)
    total = 0
    for frame in range(3):
        R_c2v, eye = _random_pose(rng, inside=(seed % 3 == 2))
        spheres = [(rng.uniform(0.5, 2.5, 3), float(rng.uniform(0.2, 0.9))) for _ in range(3)]
        planes = [((0, 0, float(rng.uniform(2.0, 3.5))), (0, 0, 1)), ((0, float(rng.uniform(2.0, 3.2)), 0), (0, 1, 0))]
        depth = synth.render_depth(R_c2v, eye, planes=planes, spheres=spheres, intr=intr, rows=rows, cols=cols,
                                   noise_mm=float(seed % 2), seed=seed * 10 + frame)
        R, t = synth.vol2cam(R_c2v, eye)
        dists = oracle.compute_dists(depth, intr)
        _, upd = oracle.integrate(want, dims, vs, tr, synth.MAX_WEIGHT, dists, R, t, intr,
                                  nthreads=oracle.max_threads())
        total += upd
        got = gpu_integrate(gpu_ctx, got, dims, vs, tr, synth.MAX_WEIGHT, dists, R, t, intr)
        assert_parity(got, want)
    assert total > 0


def test_integrate_group_test_switch_does_not_change_results(gpu_ctx):
    depth, R, t = synth.scene_s2(123, noise_mm=1.0)
    dists = oracle.compute_dists(depth, synth.INTR)
    dims, vs, tr = synth.volume_geometry(96)
    a = gpu_integrate(gpu_ctx, np.zeros(96 ** 3, dtype=np.uint32), dims, vs, tr, 64, dists, R, t, synth.INTR, frames=2)
    gpu_ctx.set_option("group_test", 0)
    try:
        b = gpu_integrate(gpu_ctx, np.zeros(96 ** 3, dtype=np.uint32), dims, vs, tr, 64, dists, R, t, synth.INTR,
                          frames=2)
    finally:
        gpu_ctx.set_option("group_test", 1)
    assert np.array_equal(a, b)
    want = np.zeros(96 ** 3, dtype=np.uint32)
    for _ in range(2):
        oracle.integrate(want, dims, vs, tr, 64, dists, R, t, synth.INTR, nthreads=oracle.max_threads())
    assert_parity(a, want)


@pytest.mark.parametrize("n,z0,z1,tile", [(1024, 500, 532, 0), (2048, 100, 108, 0), (1024, 37, 61, 4)])
def test_integrate_slab_of_a_large_volume(gpu_ctx, n, z0, z1, tile):
    """Configs C4 / C5 by their index ranges: a slab of a 1024^3 / 2048^3 volume (the whole
    volumes are 4 and 32 GiB) -- 16-MiB layers, starts deep inside the volume (hundreds of replayed z
    steps), 64-bit slab addressing -- word for word against the oracle's same slab."""
    dims, vs, tr = synth.volume_geometry(n)
    depth, R, t = synth.scene_s2(200)
    dists = oracle.compute_dists(depth, synth.INTR)
    start = np.zeros(n * n * (z1 - z0), dtype=np.uint32)
    want = start.copy()
    _, updated = oracle.integrate(want, dims, vs, tr, synth.MAX_WEIGHT, dists, R, t, synth.INTR, z0=z0, z1=z1, nthreads=16)
    assert updated > 100000
    gpu_ctx.set_option("tile", tile)
    try:
        got = gpu_integrate(gpu_ctx, start, dims, vs, tr, synth.MAX_WEIGHT, dists, R, t, synth.INTR, z0=z0, z1=z1)
    finally:
        gpu_ctx.set_option("tile", 0)
    assert np.array_equal(got >> 16, want >> 16) and np.array_equal(got, want)


@pytest.mark.parametrize("view,z0,z1", [(10, 64, 90), (100, 1500, 1530), (205, 700, 741), (290, 1940, 2000)])
def test_integrate_deep_slab_starts_from_closed_form_column_positions(gpu_ctx, view, z0, z1):
    """A slab at z0 >= 64 starts every column from the position checkpoint_kernel computes in closed form
    (yak_amd/csrc/ffwd.hpp) instead of replaying z0 rounded additions: same voxels as the oracle, which
    replays them one by one, and as the library with the shortcut switched off."""
    dims = (48, 40, 2000)
    vs, tr = (3.0 / 48, 3.0 / 40, 3.0 / 2000), 0.04
    depth, R, t = synth.scene_s2(view, noise_mm=1.0)
    dists = oracle.compute_dists(depth, synth.INTR)
    start = np.zeros(dims[0] * dims[1] * (z1 - z0), dtype=np.uint32)
    want = start.copy()
    for _ in range(2):
        oracle.integrate(want, dims, vs, tr, synth.MAX_WEIGHT, dists, R, t, synth.INTR, z0=z0, z1=z1, nthreads=8)
    assert (want != 0).sum() > 1000
    got = gpu_integrate(gpu_ctx, start, dims, vs, tr, synth.MAX_WEIGHT, dists, R, t, synth.INTR, z0=z0, z1=z1, frames=2)
    assert np.array_equal(got, want)
    gpu_ctx.set_option("checkpoint", 0)
    try:
        plain = gpu_integrate(gpu_ctx, start, dims, vs, tr, synth.MAX_WEIGHT, dists, R, t, synth.INTR, z0=z0, z1=z1, frames=2)
    finally:
        gpu_ctx.set_option("checkpoint", 1)
    assert np.array_equal(plain, want)


@pytest.mark.parametrize("tile", [3, 4, 5, 6])
def test_integrate_wave_footprint_does_not_change_results(gpu_ctx, tile):
    # "tile" folds a wave into 2^tile lanes along x by 64 >> tile rows; volumes it does not divide
    # (the second case) quietly use the next footprint that fits
    for n in ((128, 64, 40), (96, 48, 24)):
        dims, vs, tr = synth.volume_geometry(n)
        depth, R, t = synth.scene_s2(77)
        dists = oracle.compute_dists(depth, synth.INTR)
        want = np.zeros(int(np.prod(dims)), dtype=np.uint32)
        for _ in range(2):
            oracle.integrate(want, dims, vs, tr, synth.MAX_WEIGHT, dists, R, t, synth.INTR, nthreads=8)
        gpu_ctx.set_option("tile", tile)
        try:
            got = gpu_integrate(gpu_ctx, np.zeros_like(want), dims, vs, tr, synth.MAX_WEIGHT, dists, R, t, synth.INTR, frames=2)
        finally:
            gpu_ctx.set_option("tile", 0)
        assert np.array_equal(got, want)
    with pytest.raises(capi.YakHipError):
        gpu_ctx.set_option("tile", 2)


# ---------------------------------------------------------------------------- whole-brick test
def _punch_holes(depth, rng, n):
    """Blocks of pixels without a return (Dp == 0, tsdf_volume.cu:89): aligned to the brick test's 16-pixel
    cells and not, from a few pixels to a quarter of the image."""
    rows, cols = depth.shape
    for k in range(n):
        h, w = int(rng.integers(3, rows // 2)), int(rng.integers(3, cols // 2))
        y, x = int(rng.integers(0, rows - h)), int(rng.integers(0, cols - w))
        if k % 2 == 0:
            y, x, h, w = y & ~15, x & ~15, (h + 15) & ~15, (w + 15) & ~15
        depth[y:y + h, x:x + w] = 0
    return depth


@pytest.mark.parametrize("seed", range(9))
def test_integrate_brick_test_borders_near_plane_and_holes(gpu_ctx, seed):
    """The whole-brick test (integrate.hip, classify_brick) must only ever skip bricks that cannot update and
    only stream bricks that update everywhere with tsdf 1.  Cameras whose frustum cuts through the volume
    (bricks straddle the image borders), cameras inside it (bricks straddle the camera plane, vc.z <= 0 on
    one side), images with holes; every wave footprint, short chunks so that a small volume has many bricks.
    Word for word against the oracle, with the test on and off."""
    rng = np.random.default_rng(7000 + seed)
    rows, cols = [(480, 640), (240, 320), (131, 97)][seed % 3]
    f = cols * rng.uniform(0.8, 2.5)  # up to a narrow field of view: most of the volume is outside the image
    intr = (float(f), float(f * rng.uniform(0.95, 1.05)), float(cols / 2 + rng.uniform(-30, 30)),
            float(rows / 2 + rng.uniform(-30, 30)))
    dims = [(256, 64, 96), (128, 96, 80), (512, 32, 48)][seed % 3]
    _, vs, tr = synth.volume_geometry(dims)
    nvox = int(np.prod(dims))
    want = np.zeros(nvox, dtype=np.uint32)
    frames = []
    for frame in range(2):
        R_c2v, eye = _random_pose(rng, inside=(seed % 2 == 1))
        spheres = [(rng.uniform(0.5, 2.5, 3), float(rng.uniform(0.2, 0.9))) for _ in range(2)]
        planes = [((0, 0, float(rng.uniform(1.5, 3.5))), (0, 0, 1)), ((0, float(rng.uniform(1.0, 3.2)), 0), (0, 1, 0))]
        depth = synth.render_depth(R_c2v, eye, planes=planes, spheres=spheres, intr=intr, rows=rows, cols=cols,
                                   noise_mm=1.0, seed=seed * 10 + frame)
        depth = _punch_holes(depth, rng, 6)
        R, t = synth.vol2cam(R_c2v, eye)
        dists = oracle.compute_dists(depth, intr)
        oracle.integrate(want, dims, vs, tr, synth.MAX_WEIGHT, dists, R, t, intr, nthreads=oracle.max_threads())
        frames.append((dists, R, t))
    assert np.count_nonzero(want) > 0

    def run():
        got = np.zeros(nvox, dtype=np.uint32)
        for dists, R, t in frames:
            got = gpu_integrate(gpu_ctx, got, dims, vs, tr, synth.MAX_WEIGHT, dists, R, t, intr)
        return got

    stats = {}
    try:
        gpu_ctx.set_option("brick_stats", 1)
        for z_chunk, tile, side in [(16, 0, 1), (8, 3, 1), (16, 5, 1), (16, 3, 0), (0, 6, 1)]:
            gpu_ctx.set_option("z_chunk", z_chunk)
            gpu_ctx.set_option("tile", tile)
            gpu_ctx.set_option("tile_side", side)
            assert_parity(run(), want)
            s = gpu_ctx.brick_stats()
            for k in s:
                stats[k] = stats.get(k, 0) + s[k]
        # the table of z-chunks (integrate.hip chunk_table): order, shortest chunk, growth
        gpu_ctx.set_option("tile", 0)
        for z_chunk, order, cmin, ramp in [(0, 0, 0, 0), (0, 1, 2, 2), (24, 3, 4, 8), (6, 3, 0, -1), (0, -1, 6, 3)]:
            for key, v in [("z_chunk", z_chunk), ("chunk_order", order), ("chunk_min", cmin), ("chunk_ramp", ramp)]:
                gpu_ctx.set_option(key, v)
            assert_parity(run(), want)
        gpu_ctx.set_option("brick_test", 0)
        assert_parity(run(), want)
        assert sum(gpu_ctx.brick_stats().values()) > 0  # (all counted as mixed)
    finally:
        for key, v in [("brick_test", 1), ("brick_stats", 0), ("z_chunk", 0), ("tile", 0), ("tile_side", 1),
                       ("chunk_order", -1), ("chunk_min", 0), ("chunk_ramp", 0)]:
            gpu_ctx.set_option(key, v)
    assert stats["dead"] > 0 and stats["mixed"] > 0, stats  # the cases do reach the outcomes they are for


def test_integrate_brick_test_classes_on_known_scenes(gpu_ctx):
    """S0 (every voxel in view, in front of the wall): every brick is 'front' and the volume is (1.0, k) after
    k frames.  A frame without any return: every brick is 'dead' and nothing is written."""
    n = 128
    dims, vs, tr = synth.volume_geometry(n)
    depth, R, t = synth.scene_s0()
    dists = oracle.compute_dists(depth, synth.INTR)
    gpu_ctx.set_option("brick_stats", 1)
    try:
        got = gpu_integrate(gpu_ctx, np.zeros(n ** 3, dtype=np.uint32), dims, vs, tr, synth.MAX_WEIGHT, dists, R, t,
                            synth.INTR, frames=3)
        s0 = gpu_ctx.brick_stats()
        none = gpu_integrate(gpu_ctx, np.zeros(n ** 3, dtype=np.uint32), dims, vs, tr, synth.MAX_WEIGHT,
                             np.zeros_like(dists), R, t, synth.INTR)
        s1 = gpu_ctx.brick_stats()
    finally:
        gpu_ctx.set_option("brick_stats", 0)
    assert np.all(got == ((3 << 16) | 0x3C00))
    assert s0["front"] > 0 and s0["dead"] == 0 and s0["mixed"] == 0, s0
    assert not none.any()
    assert s1["dead"] > 0 and s1["front"] == 0 and s1["mixed"] == 0, s1


DEGENERATE = {
    "scaled rotation": lambda R, t, i: (1.5 * R, t, i),
    "mirrored": lambda R, t, i: (R * np.array([[-1.0], [1.0], [1.0]], dtype=np.float32), t * np.array([-1, 1, 1], dtype=np.float32), i),
    "negative fx": lambda R, t, i: (R, t, (-i[0], i[1], i[2], i[3])),
    "negative fy": lambda R, t, i: (R, t, (i[0], -i[1], i[2], i[3])),
    "NaN translation": lambda R, t, i: (R, np.array([t[0], np.nan, t[2]], dtype=np.float32), i),
    "NaN rotation entry": lambda R, t, i: (np.where(np.eye(3) > 0, R, np.float32(np.nan)).astype(np.float32), t, i),
    "infinite translation": lambda R, t, i: (R, np.array([t[0], t[1], np.inf], dtype=np.float32), i),
    "camera a thousand kilometres away": lambda R, t, i: (R, t + np.float32(1e6), i),
    "zero rotation": lambda R, t, i: (np.zeros((3, 3), dtype=np.float32), np.array([0.1, -0.05, 1.7], dtype=np.float32), i),
    "huge focal length": lambda R, t, i: (R, t, (1e7, 1e7, i[2], i[3])),
    "tiny focal length": lambda R, t, i: (R, t, (1e-3, 1e-3, i[2], i[3])),
    "principal point far outside": lambda R, t, i: (R, t, (i[0], i[1], -5000.0, 9000.0)),
    "NaN principal point": lambda R, t, i: (R, t, (i[0], i[1], float("nan"), i[3])),
}


@pytest.mark.parametrize("case", sorted(DEGENERATE))
def test_integrate_degenerate_poses_and_intrinsics(gpu_ctx, case):
    """Poses and intrinsics no camera has: the culling tests switch themselves off where their assumptions fail
    (finite pose, positive focal lengths) and the per-voxel path must still do what tsdf_volume.cu:51-110 does with
    such numbers -- the oracle's word for it, voxel word for voxel word, on top of an ordinary frame."""
    depth, R, t = synth.scene_s2(40)
    dists = oracle.compute_dists(depth, synth.INTR)
    dims, vs, tr = synth.volume_geometry((48, 40, 56))
    want = np.zeros(48 * 40 * 56, dtype=np.uint32)
    oracle.integrate(want, dims, vs, tr, synth.MAX_WEIGHT, dists, R, t, synth.INTR, nthreads=oracle.max_threads())
    got = gpu_integrate(gpu_ctx, np.zeros_like(want), dims, vs, tr, synth.MAX_WEIGHT, dists, R, t, synth.INTR)
    assert_parity(got, want)
    R2, t2, intr2 = DEGENERATE[case](np.asarray(R, dtype=np.float32).reshape(3, 3), np.asarray(t, dtype=np.float32), synth.INTR)
    oracle.integrate(want, dims, vs, tr, synth.MAX_WEIGHT, dists, R2, t2, intr2, nthreads=oracle.max_threads())
    got = gpu_integrate(gpu_ctx, got, dims, vs, tr, synth.MAX_WEIGHT, dists, R2, t2, intr2)
    assert_parity(got, want)


@pytest.mark.parametrize("rows,cols", [(1080, 1920), (1, 1), (2, 3), (9, 700), (1200, 17)])
def test_integrate_image_sizes_from_one_pixel_to_full_hd(gpu_ctx, rows, cols):
    """Images far from 640 x 480: the prepass picks coarser cells for a large image and the culling planes of a tiny
    one hold a single cell."""
    rng = np.random.default_rng(rows * 7 + cols)
    intr = (0.8 * max(cols, 8), 0.8 * max(cols, 8), cols / 2 - 0.5, rows / 2 - 0.5)
    v, u = np.mgrid[0:rows, 0:cols].astype(np.float32)
    depth = 2100.0 + 300.0 * np.sin(u / max(cols, 2) * 6.0) + 200.0 * np.cos(v / max(rows, 2) * 5.0) + rng.normal(0, 2, (rows, cols))
    depth[rng.random((rows, cols)) < 0.03] = 0
    depth = np.clip(depth, 0, 65535).astype(np.uint16)
    dists = oracle.compute_dists(depth, intr)
    dims, vs, tr = synth.volume_geometry((56, 44, 48))
    ex, ey = dims[0] * vs[0], dims[1] * vs[1]
    R = np.eye(3, dtype=np.float32)
    t = np.array([-ex / 2, -ey / 2, 0.8], dtype=np.float32)
    want = np.zeros(56 * 44 * 48, dtype=np.uint32)
    got = np.zeros_like(want)
    for _ in range(2):
        _, updated = oracle.integrate(want, dims, vs, tr, synth.MAX_WEIGHT, dists, R, t, intr, nthreads=oracle.max_threads())
        got = gpu_integrate(gpu_ctx, got, dims, vs, tr, synth.MAX_WEIGHT, dists, R, t, intr)
        assert_parity(got, want)
    assert updated > 0


@pytest.mark.parametrize("kind", ["finite halves, any weight", "any 32-bit word"])
def test_integrate_over_a_volume_of_arbitrary_words(gpu_ctx, kind):
    """A volume that did not come from integrate (uploadTSDF takes any words): weights above max_weight and, in the
    second case, infinite and NaN halves.  The running average follows the oracle's arithmetic word for word where the
    stored tsdf is finite; a NaN stays a NaN (its payload bits are the conversion's business and not compared)."""
    rng = np.random.default_rng(11)
    dims, vs, tr = synth.volume_geometry((48, 40, 56))
    n = 48 * 40 * 56
    words = rng.integers(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32)
    if kind == "finite halves, any weight":
        h = (words & 0xFFFF).astype(np.uint16)
        h = np.where((h & 0x7C00) == 0x7C00, h & 0xBBFF, h)  # clear an exponent bit of inf / NaN codes
        words = (words & 0xFFFF0000) | h
    depth, R, t = synth.scene_s2(40)
    dists = oracle.compute_dists(depth, synth.INTR)
    want = words.copy()
    got = words.copy()
    for _ in range(2):
        oracle.integrate(want, dims, vs, tr, synth.MAX_WEIGHT, dists, R, t, synth.INTR, nthreads=oracle.max_threads())
        got = gpu_integrate(gpu_ctx, got, dims, vs, tr, synth.MAX_WEIGHT, dists, R, t, synth.INTR)
    assert np.count_nonzero(want != words) > 1000
    assert np.array_equal(got >> 16, want >> 16), "weights differ"
    hw, hg = (want & 0xFFFF).astype(np.uint16), (got & 0xFFFF).astype(np.uint16)
    nan_w, nan_g = (hw & 0x7FFF) > 0x7C00, (hg & 0x7FFF) > 0x7C00
    assert np.array_equal(nan_w, nan_g), "NaN voxels differ"
    assert np.array_equal(hg[~nan_w], hw[~nan_w]), "%d finite tsdf values differ" % np.count_nonzero(hg[~nan_w] != hw[~nan_w])
    if kind == "finite halves, any weight":
        assert not nan_w.any() and np.array_equal(got, want)


@pytest.mark.parametrize("intr", [(0.0, 525.0, 319.5, 239.5), (525.0, -525.0, 319.5, 239.5), (float("nan"), 525.0, 319.5, 239.5),
                                  (525.0, 525.0, float("inf"), 239.5), (1e-30, 1e-30, 0.0, 0.0), (1e30, 1e30, 1e30, -1e30)])
def test_compute_dists_degenerate_intrinsics(gpu_ctx, intr):
    """Division by a zero focal length, NaN and infinite entries, overflow of the squared pixel offset: the half bit
    patterns of imgproc.cu:259-283 in IEEE arithmetic, as the oracle computes them (NaN codes compared as NaN)."""
    rng = np.random.default_rng(17)
    depth = rng.integers(0, 65536, size=(37, 53), dtype=np.uint16)
    depth[rng.random((37, 53)) < 0.2] = 0
    want = oracle.compute_dists(depth, intr)
    got = gpu_compute_dists(gpu_ctx, depth, intr)
    nan = (want & 0x7FFF) > 0x7C00
    assert np.array_equal((got & 0x7FFF) > 0x7C00, nan)
    assert np.array_equal(got[~nan], want[~nan])
