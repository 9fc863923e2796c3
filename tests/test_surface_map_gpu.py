"""The surface map (include/yakhip.h: yakhip_volume.surface_map): one bit per 16^3 brick of a slab, set by the
integrate kernels where a voxel may have turned negative, consulted by the ray cast to leave out the fetches of
free space.  No counterpart in the reference; the bars are

  * the invariant: every brick that holds a voxel with a negative tsdf has its bit set (checked against the voxels
    themselves, downloaded), whatever entry point integrated them and however the z-chunks were cut;
  * yakhip_surface_map_rebuild gives exactly those bricks;
  * a ray cast with the map is bit for bit the ray cast without it and the oracle's (tsdf_volume.cu:287-351
    restated in oracle/yak_oracle.c), whole volume and slab mode, also with a map that says "everywhere".
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

import oracle  # noqa: E402
from gpu_helpers import dev_u16_image  # noqa: E402
from test_raycast_gpu import assert_parity, cam2vol  # noqa: E402
from yak_amd import capi, synth  # noqa: E402
from yak_amd.shard import held_range, slab_range  # noqa: E402

pytestmark = pytest.mark.gpu

BRICK = 16


def pack_bits(flags):
    bits = np.zeros(-(-flags.size // 32) * 32, dtype=np.uint64)
    bits[:flags.size] = flags.reshape(-1)
    return (bits.reshape(-1, 32) << np.arange(32, dtype=np.uint64)).sum(axis=1).astype(np.uint32)


def unpack_bits(words, shape):
    n = int(np.prod(shape))
    return ((words[:, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(-1)[:n].astype(bool).reshape(shape)


def brick_grid(dims, layers):
    return -(-layers // BRICK), -(-dims[1] // BRICK), -(-dims[0] // BRICK)       # (bz, by, bx)


def cells_of(bricks):
    """Second level as the library defines it: a cell of 4 x 4 x 4 bricks is marked when a marked brick lies in it
    or touches it from next door (i.e. within one brick of it)."""
    bz, by, bx = bricks.shape
    grown = np.zeros((bz + 2, by + 2, bx + 2), dtype=bool)
    for dz in range(3):
        for dy in range(3):
            for dx in range(3):
                grown[dz:dz + bz, dy:dy + by, dx:dx + bx] |= bricks                # a brick marks its 26 neighbours' places
    grown = grown[1:-1, 1:-1, 1:-1]
    cz, cy, cx = -(-bz // 4), -(-by // 4), -(-bx // 4)
    pad = np.zeros((cz * 4, cy * 4, cx * 4), dtype=bool)
    pad[:bz, :by, :bx] = grown
    return pad.reshape(cz, 4, cy, 4, cx, 4).any(axis=(1, 3, 5))


def negative_bricks(vol_host, dims, layers):
    """The bricks of a (layers, Y, X) slab that hold a negative tsdf, as a (bz, by, bx) array."""
    X, Y = dims[0], dims[1]
    w = vol_host.reshape(layers, Y, X)
    neg = ((w & 0x8000) != 0) & ((w & 0x7FFF) != 0)
    bz, by, bx = brick_grid(dims, layers)
    pad = np.zeros((bz * BRICK, by * BRICK, bx * BRICK), dtype=bool)
    pad[:layers, :Y, :X] = neg
    return pad.reshape(bz, BRICK, by, BRICK, bx, BRICK).any(axis=(1, 3, 5))


def whole_map(bricks):
    """Brick words followed by cell words: the layout of yakhip_volume.surface_map."""
    return np.concatenate([pack_bits(bricks), pack_bits(cells_of(bricks))])


def popcount(words):
    return int(np.unpackbits(words.view(np.uint8)).sum())


class Slab:
    """A slab on the device with its surface map."""

    def __init__(self, dims, vs, tr, z0=0, z1=None, with_map=True):
        self.dims, self.vs, self.tr = dims, vs, tr
        self.z0, self.z1 = z0, dims[2] if z1 is None else z1
        self.vol = torch.zeros(dims[0] * dims[1] * (self.z1 - self.z0), dtype=torch.int32, device="cuda")
        self.plain = capi.volume_desc(self.vol.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT, self.z0, self.z1)
        self.map = torch.full((capi.surface_map_bytes(self.plain) // 4,), -1, dtype=torch.int32, device="cuda")
        self.desc = capi.volume_desc(self.vol.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT, self.z0, self.z1,
                                     surface_map=self.map.data_ptr() if with_map else None)

    def voxels(self):
        torch.cuda.synchronize()
        return self.vol.cpu().numpy().view(np.uint32)

    def bits(self):
        torch.cuda.synchronize()
        return self.map.cpu().numpy().view(np.uint32)


def orbit_frames(views, intr=synth.INTR, rows=synth.ROWS, cols=synth.COLS):
    out = []
    for v in views:
        depth, R, t = synth.scene_s2(v, noise_mm=1.0)
        out.append((oracle.compute_dists(depth, synth.INTR), R, t))
    return out


@pytest.mark.parametrize("dims,z0,z1,how", [((128, 128, 128), 0, None, "single"), ((128, 128, 128), 0, None, "batch"),
                                            ((128, 128, 128), 0, None, "static"), ((72, 40, 100), 0, None, "single"),
                                            ((96, 96, 160), 37, 121, "single"), ((96, 96, 160), 37, 121, "batch")])
def test_integrate_marks_every_brick_that_holds_a_negative_voxel(gpu_ctx, dims, z0, z1, how):
    _, vs, tr = synth.volume_geometry(dims)
    s = Slab(dims, vs, tr, z0, z1)
    gpu_ctx.clear_volume(s.desc)
    assert not s.bits().any()                                     # clear clears the map with the voxels
    frames = orbit_frames([0, 40, 95, 150, 222, 260])
    dev = [dev_u16_image(d) for d, _, _ in frames]
    want = np.zeros(s.vol.numel(), dtype=np.uint32)
    for d, R, t in frames:
        oracle.integrate(want, dims, vs, tr, synth.MAX_WEIGHT, d, R, t, synth.INTR, z0=s.z0, z1=s.z1, nthreads=8)
    rows, cols = frames[0][0].shape
    args = [(dd.data_ptr(), cols * 2, rows, cols, R.reshape(-1), t, synth.INTR) for dd, (_, R, t) in zip(dev, frames)]
    try:
        if how == "batch":
            gpu_ctx.integrate_batch(s.desc, args[:4])
            gpu_ctx.integrate_batch(s.desc, args[4:])
        else:
            for i, a in enumerate(args):
                # every way of cutting the z-chunks: the marks are flushed where a walk leaves a brick's layers
                gpu_ctx.set_option("z_chunk", [0, 7, 16, 5, 24, 1][i])
                (gpu_ctx.integrate_static if how == "static" else gpu_ctx.integrate)(s.desc, *a)
    finally:
        gpu_ctx.set_option("z_chunk", 0)
    assert np.array_equal(s.voxels(), want)                       # (the voxels themselves: word for word, as ever)
    exact = negative_bricks(want, dims, s.z1 - s.z0)
    n_exact = int(exact.sum())
    got = s.bits().copy()
    n_words = pack_bits(exact).size
    marked = unpack_bits(got[:n_words], exact.shape)
    assert n_exact > 10
    assert not (exact & ~marked).any(), "a brick with a negative voxel is not marked"
    # marks follow the truncation band, not the frustum (a brick stays marked when later frames, seeing it from the
    # other side, lift its voxels back above zero: hence the factor)
    assert int(marked.sum()) <= 3 * n_exact + 8
    # second level: every cell within one brick of a marked brick (the kernels mark the cells of every brick layer
    # a walk touched, for the lanes that marked a brick during it: a superset, bounded by the chunk length)
    cells = unpack_bits(got[n_words:], cells_of(marked).shape)
    assert not (cells_of(marked) & ~cells).any(), "a cell next to a marked brick is not marked"
    grown = np.zeros_like(marked)
    for dz in range(-4, 5):                                        # four brick layers either way cover any chunk here
        lo, hi = max(0, dz), marked.shape[0] + min(0, dz)
        grown[lo:hi] |= marked[lo - dz:hi - dz]
    assert not (cells & ~cells_of(grown)).any()
    gpu_ctx.surface_map_rebuild(s.desc)
    assert np.array_equal(s.bits(), whole_map(exact))


def raycast(ctx, desc, R, t, Rinv, intr, rows, cols, depth=False):
    nrm = torch.full((rows, cols, 4), 7.0, dtype=torch.float32, device="cuda")
    if depth:
        img = torch.full((rows, cols), 7, dtype=torch.int16, device="cuda")
        ctx.raycast_depth(desc, R.reshape(-1), t, Rinv.reshape(-1), intr, img.data_ptr(), cols * 2, nrm.data_ptr(), cols * 16,
                          rows, cols)
    else:
        img = torch.full((rows, cols, 4), 7.0, dtype=torch.float32, device="cuda")
        ctx.raycast_points(desc, R.reshape(-1), t, Rinv.reshape(-1), intr, img.data_ptr(), cols * 16, nrm.data_ptr(),
                           cols * 16, rows, cols)
    torch.cuda.synchronize()
    return img.cpu().numpy(), nrm.cpu().numpy()


def same_bits(a, b):
    return all(np.array_equal(x.view(np.uint8), y.view(np.uint8)) for x, y in zip(a, b))


@pytest.mark.parametrize("dims", [(128, 128, 128), (72, 40, 100)])
def test_raycast_with_the_map_is_the_raycast_without_it_and_the_oracles(gpu_ctx, dims):
    _, vs, tr = synth.volume_geometry(dims)
    s = Slab(dims, vs, tr)
    gpu_ctx.clear_volume(s.desc)
    rows, cols = synth.ROWS, synth.COLS
    for d, R, t in orbit_frames([0, 10, 20, 30, 40, 50, 120, 250]):
        dd = dev_u16_image(d)
        gpu_ctx.integrate(s.desc, dd.data_ptr(), cols * 2, rows, cols, R.reshape(-1), t, synth.INTR)
    vol = s.voxels().copy()
    kept = s.map.clone()
    total = 0
    for view in (5, 47, 133, 260):
        _, R_v2c, t_v2c = synth.scene_s2(view)
        R, t = cam2vol(R_v2c, t_v2c)
        want_p, want_n, _, hits = oracle.raycast(vol, dims, vs, tr, R, t, R_v2c, synth.INTR, rows, cols, nthreads=8)
        total += hits
        plain = raycast(gpu_ctx, s.plain, R, t, R_v2c, synth.INTR, rows, cols)
        assert_parity(plain[0], plain[1], want_p, want_n)
        s.map.copy_(kept)                                         # as integrate left it
        assert same_bits(raycast(gpu_ctx, s.desc, R, t, R_v2c, synth.INTR, rows, cols), plain)
        gpu_ctx.surface_map_rebuild(s.desc)                       # exactly the bricks with a negative voxel
        assert same_bits(raycast(gpu_ctx, s.desc, R, t, R_v2c, synth.INTR, rows, cols), plain)
        s.map.fill_(-1)                                           # a map that says "everywhere": nothing is skipped
        assert same_bits(raycast(gpu_ctx, s.desc, R, t, R_v2c, synth.INTR, rows, cols), plain)
        s.map.copy_(kept)
        gpu_ctx.set_option("surface_skip", 0)                     # the switch
        try:
            assert same_bits(raycast(gpu_ctx, s.desc, R, t, R_v2c, synth.INTR, rows, cols), plain)
        finally:
            gpu_ctx.set_option("surface_skip", 1)
        dplain = raycast(gpu_ctx, s.plain, R, t, R_v2c, synth.INTR, rows, cols, depth=True)
        assert same_bits(raycast(gpu_ctx, s.desc, R, t, R_v2c, synth.INTR, rows, cols, depth=True), dplain)
    assert total > 4000


def test_raycast_of_an_uploaded_volume_after_rebuild_and_cameras_inside(gpu_ctx):
    """Voxels written behind the library's back (here: uploaded), then yakhip_surface_map_rebuild; cameras inside
    the volume, looking along every axis, so that marches start in the middle of bricks."""
    dims, vs, tr = synth.volume_geometry(96)
    vol = np.zeros(96 ** 3, dtype=np.uint32)
    for f in (0, 60, 120, 180, 240):
        depth, R, t = synth.scene_s2(f)
        oracle.integrate(vol, dims, vs, tr, synth.MAX_WEIGHT, oracle.compute_dists(depth, synth.INTR), R, t, synth.INTR, nthreads=8)
    s = Slab(dims, vs, tr)
    s.vol.copy_(torch.from_numpy(vol.view(np.int32)).cuda())
    gpu_ctx.surface_map_rebuild(s.desc)
    exact = negative_bricks(vol, dims, 96)
    assert exact.sum() > 10 and np.array_equal(s.bits(), whole_map(exact))
    rows, cols, intr = 120, 160, (145.0, 145.0, 79.5, 59.5)
    rng = np.random.default_rng(11)
    hits = 0
    for k in range(8):
        eye = np.array([1.5, 1.2, 1.5]) + rng.uniform(-0.6, 0.6, 3)
        target = eye + np.eye(3)[k % 3] * (1 if k < 4 else -1) + rng.uniform(-0.3, 0.3, 3)
        R_c2v, _ = synth.look_at(eye, target)
        R = np.ascontiguousarray(R_c2v, dtype=np.float32)
        t = eye.astype(np.float32)
        Rinv = np.ascontiguousarray(R.T)
        want_p, want_n, _, h = oracle.raycast(vol, dims, vs, tr, R, t, Rinv, intr, rows, cols, nthreads=8)
        hits += h
        got = raycast(gpu_ctx, s.desc, R, t, Rinv, intr, rows, cols)
        assert_parity(got[0], got[1], want_p, want_n)
    assert hits > 5000


@pytest.mark.parametrize("world", [2, 3])
def test_slab_raycast_with_the_map_is_the_slab_raycast_without_it(gpu_ctx, world):
    """Each "rank" integrates the layers it holds (own + halo) with its own map, then marches with and without it."""
    dims, vs, tr = synth.volume_geometry(96)
    rows, cols, intr = 120, 160, (145.0, 145.0, 79.5, 59.5)
    halo = capi.raycast_halo(capi.volume_desc(256, dims, vs, tr, synth.MAX_WEIGHT))
    frames = orbit_frames([0, 60, 120, 180, 240])
    dev = [dev_u16_image(d) for d, _, _ in frames]
    r_, c_ = frames[0][0].shape
    for rank in range(world):
        own = slab_range(dims[2], world, rank)
        h0, h1 = held_range(dims[2], own[0], own[1], halo)
        s = Slab(dims, vs, tr, h0, h1)
        gpu_ctx.clear_volume(s.desc)
        for dd, (_, R, t) in zip(dev, frames):
            gpu_ctx.integrate(s.desc, dd.data_ptr(), c_ * 2, r_, c_, R.reshape(-1), t, synth.INTR)
        exact = negative_bricks(s.voxels(), dims, h1 - h0)
        assert exact.sum() > 5 and not (whole_map(exact) & ~s.bits()).any()
        for view in (10, 130, 250):
            _, R_v2c, t_v2c = synth.scene_s2(view)
            R, t = cam2vol(R_v2c, t_v2c)
            Rinv = np.ascontiguousarray(R.T)
            outs = []
            for desc in (s.plain, s.desc):
                keys = torch.full((rows, cols), -5, dtype=torch.int32, device="cuda")
                nrm = torch.full((rows, cols, 4), 7.0, dtype=torch.float32, device="cuda")
                img = torch.full((rows, cols, 4), 7.0, dtype=torch.float32, device="cuda")
                gpu_ctx.raycast_slab(desc, own[0], own[1], R.reshape(-1), t, Rinv.reshape(-1), intr, keys.data_ptr(), cols * 4,
                                     nrm.data_ptr(), cols * 16, rows, cols, points_ptr=img.data_ptr(), points_pitch=cols * 16)
                torch.cuda.synchronize()
                outs.append((keys.cpu().numpy(), img.cpu().numpy(), nrm.cpu().numpy()))
            assert (outs[0][0] != 0x7FFFFFFF).sum() > 100
            assert same_bits(outs[0], outs[1])


def test_raycast_512_with_the_map_on_a_volume_fused_on_the_gpu(gpu_ctx):
    """Full size (BASELINE configs[1] volume): batched and single-frame integration of the orbit, then views along the
    z axis and oblique ones, with the map against without."""
    dims, vs, tr = synth.volume_geometry(512)
    s = Slab(dims, vs, tr)
    gpu_ctx.clear_volume(s.desc)
    rows, cols = synth.ROWS, synth.COLS
    frames = orbit_frames(list(range(0, 300, 25)))
    dev = [dev_u16_image(d) for d, _, _ in frames]
    args = [(dd.data_ptr(), cols * 2, rows, cols, R.reshape(-1), t, synth.INTR) for dd, (_, R, t) in zip(dev, frames)]
    gpu_ctx.integrate_batch(s.desc, args[:8])
    for a in args[8:]:
        gpu_ctx.integrate(s.desc, *a)
    got = s.bits().copy()
    gpu_ctx.surface_map_rebuild(s.desc)
    exact = s.bits().copy()
    assert popcount(exact) > 1000 and not (exact & ~got).any() and popcount(got) <= 2 * popcount(exact)
    s.map.copy_(torch.from_numpy(got.view(np.int32)).cuda())
    for view in (0, 37, 100, 222):
        _, R_v2c, t_v2c = synth.scene_s2(view)
        R, t = cam2vol(R_v2c, t_v2c)
        plain = raycast(gpu_ctx, s.plain, R, t, R_v2c, synth.INTR, rows, cols)
        assert (~np.isnan(plain[0][..., 0])).sum() > 50000
        assert same_bits(raycast(gpu_ctx, s.desc, R, t, R_v2c, synth.INTR, rows, cols), plain)


@pytest.mark.parametrize("n,layers", [(1024, 528), (1024, 1024)])
def test_raycast_with_the_map_on_volumes_of_2_gib_and_more(gpu_ctx, n, layers):
    """Held layers of 2 GiB and more do not fit the buffer descriptor the samples of smaller volumes are read through
    (raycast.hip, fetch_tsdf_skip<..., BIG>), and a 1024^3 volume's map is 8320 words of LDS: the ray cast with the map
    must still be the ray cast without it, bit for bit (which the oracle pins at the sizes it finishes)."""
    dims, vs, tr = synth.volume_geometry(n)
    dims = (dims[0], dims[1], layers)
    s = Slab(dims, vs, tr)
    assert int(np.prod(dims)) * 4 >= 2 ** 31
    gpu_ctx.clear_volume(s.desc)
    rows, cols = synth.ROWS, synth.COLS
    frames = orbit_frames(list(range(0, 300, 50)))
    dev = [dev_u16_image(d) for d, _, _ in frames]
    for dd, (_, R, t) in zip(dev, frames):
        gpu_ctx.integrate(s.desc, dd.data_ptr(), cols * 2, rows, cols, R.reshape(-1), t, synth.INTR)
    for view in (0, 100, 222):
        _, R_v2c, t_v2c = synth.scene_s2(view)
        R, t = cam2vol(R_v2c, t_v2c)
        plain = raycast(gpu_ctx, s.plain, R, t, R_v2c, synth.INTR, rows, cols)
        assert (~np.isnan(plain[0][..., 0])).sum() > 20000
        assert same_bits(raycast(gpu_ctx, s.desc, R, t, R_v2c, synth.INTR, rows, cols), plain)
        assert same_bits(raycast(gpu_ctx, s.desc, R, t, R_v2c, synth.INTR, rows, cols, depth=True),
                         raycast(gpu_ctx, s.plain, R, t, R_v2c, synth.INTR, rows, cols, depth=True))
    del s
    torch.cuda.empty_cache()


def test_marching_cubes_and_cloud_with_the_map_give_the_same_output(gpu_ctx):
    """The count pass leaves out what no marked brick is near; the triangle list is the same, entry for entry --
    whole slab, a slab that starts inside the volume, and a layer range of a slab (yakhip_marching_cubes_layers)."""
    for dims, z0, z1 in [((128, 128, 128), 0, None), ((72, 40, 100), 0, None), ((96, 96, 160), 37, 121)]:
        _, vs, tr = synth.volume_geometry(dims)
        s = Slab(dims, vs, tr, z0, z1)
        gpu_ctx.clear_volume(s.desc)
        rows, cols = synth.ROWS, synth.COLS
        for d, R, t in orbit_frames([0, 60, 120, 180, 240]):
            dd = dev_u16_image(d)
            gpu_ctx.integrate(s.desc, dd.data_ptr(), cols * 2, rows, cols, R.reshape(-1), t, synth.INTR)

        def mesh(call, *args):
            n = call(*args, 2, vs[0])
            tri = torch.full((max(n, 1), 9), 7.0, dtype=torch.float32, device="cuda")
            assert call(*args, 2, vs[0], tri.data_ptr(), n) == n
            return tri.cpu().numpy()[:n]

        plain = mesh(gpu_ctx.marching_cubes, s.plain)
        assert len(plain) > 500
        assert np.array_equal(mesh(gpu_ctx.marching_cubes, s.desc).view(np.uint32), plain.view(np.uint32))
        # a range of the slab's layers: what a separate slab over those layers gives
        a, b = s.z0 + 5, s.z1 - 9
        plane = dims[0] * dims[1]
        sub = capi.volume_desc(s.vol.data_ptr() + 4 * plane * (a - s.z0), dims, vs, tr, synth.MAX_WEIGHT, a, b)
        want = mesh(gpu_ctx.marching_cubes, sub)
        assert len(want) > 300
        for desc in (s.desc, s.plain):
            assert np.array_equal(mesh(gpu_ctx.marching_cubes_layers, desc, a, b).view(np.uint32), want.view(np.uint32))
        # the cloud extraction's count pass does the same (yakhip_extract_cloud)
        pose_R, pose_t = np.eye(3, dtype=np.float32).reshape(-1), np.zeros(3, dtype=np.float32)

        def cloud(desc):
            n = gpu_ctx.extract_cloud(desc, pose_R, pose_t)
            pts = torch.full((max(n, 1), 4), 7.0, dtype=torch.float32, device="cuda")
            assert gpu_ctx.extract_cloud(desc, pose_R, pose_t, pts.data_ptr(), n) == n
            return pts.cpu().numpy()[:n]

        plain_cloud = cloud(s.plain)
        assert len(plain_cloud) > 500
        assert np.array_equal(cloud(s.desc).view(np.uint32), plain_cloud.view(np.uint32))
    with pytest.raises(capi.YakHipError):
        gpu_ctx.marching_cubes_layers(s.desc, s.z0 - 1, s.z1, 2, 1.0)


def test_surface_map_arguments(gpu_ctx):
    dims, vs, tr = synth.volume_geometry((40, 33, 50))
    s = Slab(dims, vs, tr, 3, 47)
    assert capi.surface_map_bytes(s.plain) == 4 * (-(-(3 * 3 * 3) // 32) + 1)       # 27 bricks, 1 cell
    with pytest.raises(capi.YakHipError):
        gpu_ctx.surface_map_rebuild(s.plain)                      # no map to rebuild
    bad = capi.volume_desc(s.vol.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT, 3, 47, surface_map=s.map.data_ptr() + 2)
    with pytest.raises(capi.YakHipError):
        gpu_ctx.clear_volume(bad)                                 # misaligned map


def test_rebuilt_map_of_a_sparse_uploaded_volume_changes_no_output(gpu_ctx):
    """An uploaded volume that is free space (+1, weight 5) except for blobs of arbitrary finite words and single
    voxels at brick corners -- negative values, -0.0, +0.0, denormals: most bricks stay unmarked, and what is next to a
    brick boundary must still be seen.  Ray cast, marching cubes and cloud with the rebuilt map = without it."""
    rng = np.random.default_rng(31)
    dims = (96, 80, 112)
    _, vs, tr = synth.volume_geometry(dims)
    X, Y, Z = dims
    vol = np.full((Z, Y, X), (5 << 16) | 0x3C00, dtype=np.uint32)
    for _ in range(6):    # blobs of anything
        cx, cy, cz = (int(rng.integers(8, n - 8)) for n in (X, Y, Z))
        r = int(rng.integers(3, 8))
        blob = rng.integers(0, 2 ** 32, size=(2 * r, 2 * r, 2 * r), dtype=np.uint64).astype(np.uint32)
        h = (blob & 0xFFFF).astype(np.uint16)
        blob = (blob & 0xFFFF0000) | np.where((h & 0x7C00) == 0x7C00, h & 0xBBFF, h)
        vol[cz - r:cz + r, cy - r:cy + r, cx - r:cx + r] = blob
    singles = [0xBC00, 0x8000, 0x0000, 0x8001, 0x0001, 0xB800, 0x8400]           # -1, -0, +0, -denormal, +denormal, -0.5, -2^-14
    for k in range(60):   # single voxels on and next to brick corners and faces
        bx, by, bz = (int(rng.integers(1, n // 16)) * 16 for n in (X, Y, Z))
        dx, dy, dz = (int(v) for v in rng.integers(-1, 1, 3))
        vol[bz + dz, by + dy, bx + dx] = (7 << 16) | singles[k % len(singles)]
    vol = vol.reshape(-1)
    s = Slab(dims, vs, tr)
    s.vol.copy_(torch.from_numpy(vol.view(np.int32)).cuda())
    gpu_ctx.surface_map_rebuild(s.desc)
    bricks = s.bits()[:capi.surface_map_bytes(s.plain) // 4]
    n_bricks = (X // 16) * (Y // 16) * (Z // 16)
    marked = sum(bin(int(w)).count("1") for w in bricks[:(n_bricks + 31) // 32])
    assert 10 < marked < n_bricks // 2, marked

    rows, cols = 240, 320
    intr = tuple(v / 2 for v in synth.INTR)
    for view in (10, 130, 250):
        _, R_v2c, t_v2c = synth.scene_s2(view)
        R, t = cam2vol(R_v2c, t_v2c)
        plain = raycast(gpu_ctx, s.plain, R, t, R_v2c, intr, rows, cols)
        assert (~np.isnan(plain[0][..., 0])).sum() > 100
        assert same_bits(raycast(gpu_ctx, s.desc, R, t, R_v2c, intr, rows, cols), plain)

    def mesh(desc, min_weight):
        n = gpu_ctx.marching_cubes(desc, min_weight, vs[0])
        tri = torch.full((max(n, 1), 9), 7.0, dtype=torch.float32, device="cuda")
        assert gpu_ctx.marching_cubes(desc, min_weight, vs[0], tri.data_ptr(), n) == n
        return tri.cpu().numpy()[:n]

    for min_weight in (0, 6):
        plain_mesh = mesh(s.plain, min_weight)
        assert len(plain_mesh) > 200
        assert np.array_equal(mesh(s.desc, min_weight).view(np.uint32), plain_mesh.view(np.uint32))
        want = oracle.marching_cubes(vol, dims, min_weight, vs[0])
        assert len(want) == len(plain_mesh)

    pose_R, pose_t = np.eye(3, dtype=np.float32).reshape(-1), np.zeros(3, dtype=np.float32)

    def cloud(desc):
        n = gpu_ctx.extract_cloud(desc, pose_R, pose_t)
        pts = torch.full((max(n, 1), 4), 7.0, dtype=torch.float32, device="cuda")
        assert gpu_ctx.extract_cloud(desc, pose_R, pose_t, pts.data_ptr(), n) == n
        return pts.cpu().numpy()[:n]

    plain_cloud = cloud(s.plain)
    assert len(plain_cloud) > 200
    assert np.array_equal(cloud(s.desc).view(np.uint32), plain_cloud.view(np.uint32))
    assert len(oracle.extract_cloud(vol, dims, vs, np.eye(3, dtype=np.float32), pose_t)) == len(plain_cloud)
