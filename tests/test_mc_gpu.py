"""GPU parity: yakhip_marching_cubes (through the C ABI) against the CPU oracle, which the reference's
dipyramid known answer pins (tests/test_oracle_mc.py).  Bar: the same triangle SET, bit for bit
(the reference's own emission order is nondeterministic)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

import oracle  # noqa: E402
from test_oracle_mc import DIPYRAMID_VERTICES, clean, sphere_volume  # noqa: E402
from yak_amd import capi, synth  # noqa: E402

pytestmark = pytest.mark.gpu


def canon(tris):
    """Order-free form of a triangle list: rows of 9 float bit patterns, sorted."""
    rows = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1, 9).view(np.uint32)
    return rows[np.lexsort(rows.T[::-1])]


def gpu_mc(ctx, vol_dev, dims, min_weight, scale=1.0, z0=0, z1=None, slack=0):
    desc = capi.volume_desc(vol_dev.data_ptr(), dims, (1, 1, 1), 1.0, 64, z0, z1)
    n = ctx.marching_cubes(desc, min_weight, scale)                       # count only
    buf = torch.full((max(n + slack, 1), 9), 7.0, dtype=torch.float32, device="cuda")
    m = ctx.marching_cubes(desc, min_weight, scale, buf.data_ptr(), n + slack)
    assert m == n
    out = buf.cpu().numpy()
    assert (out[n:] == 7.0).all()                                         # nothing beyond the count is written
    return out[:n].reshape(-1, 3, 3)


def test_mc_dipyramid_reference_known_answer(gpu_ctx):
    vol = np.full(27, 0x00013C00, dtype=np.uint32)
    vol[13] = 0x0001BC00
    tris = gpu_mc(gpu_ctx, torch.from_numpy(vol.view(np.int32)).cuda(), (3, 3, 3), 1, 0.01, slack=3)
    assert tris.shape == (8, 3, 3)                                        # test/marching_cubes_tests.cpp:121
    verts, faces = clean(tris)
    assert len(verts) == 6 and len(faces) == 8                            # :133
    assert np.array_equal(verts, DIPYRAMID_VERTICES[np.lexsort(DIPYRAMID_VERTICES.T[::-1])])
    assert np.array_equal(canon(tris), canon(oracle.marching_cubes(vol, (3, 3, 3), 1, 0.01)))


@pytest.mark.parametrize("n,r,scale", [(64, 20.3, 1.0), (130, 51.7, 0.005859375), (256, 25.6, 0.01)])
def test_mc_sphere_c1(gpu_ctx, n, r, scale):
    vol = sphere_volume(n, r)
    want = oracle.marching_cubes(vol, (n, n, n), 4, scale)
    got = gpu_mc(gpu_ctx, torch.from_numpy(vol.view(np.int32)).cuda(), (n, n, n), 4, scale)
    assert len(got) == len(want) > 5000
    assert np.array_equal(canon(got), canon(want))


def fused(n, frames):
    dims, vs, tr = synth.volume_geometry(n)
    vol = np.zeros(int(np.prod(dims)), dtype=np.uint32)
    for f in frames:
        depth, R, t = synth.scene_s2(f)
        oracle.integrate(vol, dims, vs, tr, synth.MAX_WEIGHT, oracle.compute_dists(depth, synth.INTR), R, t, synth.INTR,
                         nthreads=8)
    return vol, dims


@pytest.mark.parametrize("min_weight", [1, 3, 6])
def test_mc_fused_scene_weights(gpu_ctx, min_weight):
    vol, dims = fused((96, 80, 112), [0, 20, 40, 60, 80, 100])
    want = oracle.marching_cubes(vol, dims, min_weight, 3.0 / 96)
    got = gpu_mc(gpu_ctx, torch.from_numpy(vol.view(np.int32)).cuda(), dims, min_weight, 3.0 / 96)
    assert len(want) > (1000 if min_weight < 6 else 0)
    assert len(got) == len(want) and np.array_equal(canon(got), canon(want))


def test_mc_slabs_with_one_layer_overlap_tile_the_volume(gpu_ctx):
    vol, dims = fused(64, [0, 75, 150, 225])
    plane = dims[0] * dims[1]
    dev = torch.from_numpy(vol.view(np.int32)).cuda()
    whole = gpu_mc(gpu_ctx, dev, dims, 1)
    parts = []
    for z0, z1 in [(0, 17), (16, 41), (40, 64)]:                          # slab k ends one layer into slab k + 1
        parts.append(gpu_mc(gpu_ctx, dev[z0 * plane:], dims, 1, z0=z0, z1=z1))
    assert sum(len(p) for p in parts) == len(whole) > 1000
    assert np.array_equal(canon(np.concatenate(parts)), canon(whole))
    assert np.array_equal(canon(whole), canon(oracle.marching_cubes(vol, dims, 1)))


def test_mc_capacity_and_degenerate_volumes(gpu_ctx):
    vol = sphere_volume(32, 9.5)
    dev = torch.from_numpy(vol.view(np.int32)).cuda()
    desc = capi.volume_desc(dev.data_ptr(), (32, 32, 32), (1, 1, 1), 1.0, 64)
    n = gpu_ctx.marching_cubes(desc, 4, 1.0)
    assert n == len(oracle.marching_cubes(vol, (32, 32, 32), 4))
    buf = torch.full((n, 9), 7.0, dtype=torch.float32, device="cuda")
    assert gpu_ctx.marching_cubes(desc, 4, 1.0, buf.data_ptr(), n - 1) == n   # too small: count comes back ...
    assert (buf == 7.0).all().item()                                            # ... and nothing is written
    # no cube fits in a single layer / row / column
    one = capi.volume_desc(dev.data_ptr(), (32, 32, 32), (1, 1, 1), 1.0, 64, 5, 6)
    assert gpu_ctx.marching_cubes(one, 1, 1.0) == 0
    flat = capi.volume_desc(dev.data_ptr(), (1024, 1, 32), (1, 1, 1), 1.0, 64)
    assert gpu_ctx.marching_cubes(flat, 1, 1.0) == 0
    assert gpu_ctx.marching_cubes(desc, 5, 1.0) == 0                            # every weight is 4
    with pytest.raises(capi.YakHipError):
        gpu_ctx.marching_cubes(desc, 4, 1.0, buf.data_ptr() + 2, n)           # misaligned output


def noise_volume(dims, seed):
    """Every cube its own case: random tsdf halves in [-1, 1] (signed zeros among them), weights 0..3."""
    rng = np.random.default_rng(seed)
    n = int(np.prod(dims))
    tsdf = rng.uniform(-1, 1, n).astype(np.float16)
    tsdf[rng.random(n) < 0.05] = np.float16(0.0)
    tsdf[rng.random(n) < 0.05] = np.float16(-0.0)
    weight = rng.integers(0, 4, n).astype(np.uint32)
    return tsdf.view(np.uint16).astype(np.uint32) | (weight << 16)


@pytest.mark.parametrize("dims,min_weight", [((70, 19, 21), 0), ((64, 8, 40), 1), ((130, 33, 18), 1), ((5, 5, 70), 1)])
def test_mc_noise_every_case_and_full_cube_queues(gpu_ctx, dims, min_weight):
    """Dense random surface: up to 5 triangles in every cube, so the emit pass's cube queue fills and
    drains many times per tile, on volumes whose edges cut the 63 x 4 wave tiles."""
    vol = noise_volume(dims, sum(dims))
    want = oracle.marching_cubes(vol, dims, min_weight, 0.37)
    got = gpu_mc(gpu_ctx, torch.from_numpy(vol.view(np.int32)).cuda(), dims, min_weight, 0.37)
    assert len(got) == len(want) > 50
    assert np.array_equal(canon(got), canon(want))


@pytest.mark.parametrize("depth", [1, 3, 5, 16, 64])
def test_mc_tile_depth_option_does_not_change_the_set(gpu_ctx, depth):
    vol = noise_volume((66, 9, 37), 11)
    vol[: 66 * 9 * 12] = 0x00013C00                                        # free space below: empty tiles
    want = canon(oracle.marching_cubes(vol, (66, 9, 37), 1))
    gpu_ctx.set_option("mc_z_chunk", depth)
    try:
        got = gpu_mc(gpu_ctx, torch.from_numpy(vol.view(np.int32)).cuda(), (66, 9, 37), 1)
    finally:
        gpu_ctx.set_option("mc_z_chunk", 0)
    assert np.array_equal(canon(got), want)


def test_mc_wide_slab_of_a_2048_volume(gpu_ctx):
    """Config 5's shape in small: the first 24 layers of a 2048 x 512 x 2048 volume as one rank's slab (33
    wave tiles per row, the last one partial; 25 M voxels), a tilted plane through it."""
    X, Y, Z1 = 2048, 512, 24
    x = np.arange(X, dtype=np.float32)[None, None, :]
    y = np.arange(Y, dtype=np.float32)[None, :, None]
    z = np.arange(Z1, dtype=np.float32)[:, None, None]
    d = np.clip((z - 11.3 + 0.004 * x - 0.007 * y) / 3.0, -1, 1).astype(np.float16)
    vol = (d.view(np.uint16).astype(np.uint32) | (5 << 16)).reshape(-1)
    dev = torch.from_numpy(vol.view(np.int32)).cuda()
    got = gpu_mc(gpu_ctx, dev, (X, Y, 2048), 4, 0.0014648, z0=0, z1=Z1)
    want = oracle.marching_cubes(vol, (X, Y, Z1), 4, 0.0014648)      # the slab as a volume of its own
    assert len(got) == len(want) > 1500000
    assert np.array_equal(canon(got), canon(want))


def test_mc_512_volume_fused_on_gpu(gpu_ctx):
    n = 512
    dims, vs, tr = synth.volume_geometry(n)
    vol = torch.zeros(n ** 3, dtype=torch.int32, device="cuda")
    desc = capi.volume_desc(vol.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT)
    d_dists = torch.empty((synth.ROWS, synth.COLS), dtype=torch.int16, device="cuda")
    for f in (0, 30, 60, 90, 120):
        depth, R, t = synth.scene_s2(f)
        d_depth = torch.from_numpy(depth.view(np.int16)).cuda()
        gpu_ctx.compute_dists(d_depth.data_ptr(), synth.COLS * 2, d_dists.data_ptr(), synth.COLS * 2, synth.ROWS,
                              synth.COLS, synth.INTR)
        gpu_ctx.integrate(desc, d_dists.data_ptr(), synth.COLS * 2, synth.ROWS, synth.COLS, R.reshape(-1), t, synth.INTR)
    torch.cuda.synchronize()
    got = gpu_mc(gpu_ctx, vol, dims, 2, vs[0])
    want = oracle.marching_cubes(vol.cpu().numpy().view(np.uint32), dims, 2, vs[0])
    assert len(got) == len(want) > 100000
    assert np.array_equal(canon(got), canon(want))


# ---- yakhip_weld_vertices: the exact point merging of MarchingCubesParameters::clean ---------------------------

def numpy_weld(tris):
    """Distinct corners in the order of their first appearance in the list, and every corner's vertex number
    (include/yakhip.h: yakhip_weld_vertices)."""
    bits = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1, 3).view(np.uint32)
    order = np.lexsort((bits[:, 2], bits[:, 1], bits[:, 0]))  # stable: equal points stay in list order
    s = bits[order]
    head = np.ones(len(s), dtype=bool)
    head[1:] = (s[1:] != s[:-1]).any(axis=1)
    group = np.cumsum(head) - 1                                # group of each sorted place
    first = order[head]                                        # first list position of every group
    rank = np.empty(len(first), dtype=np.int64)
    rank[np.argsort(first, kind="stable")] = np.arange(len(first))
    index = np.empty(len(s), dtype=np.uint32)
    index[order] = rank[group]
    return bits[np.sort(first)], index


def gpu_weld(ctx, tris):
    t = torch.from_numpy(np.ascontiguousarray(tris, dtype=np.float32).reshape(-1, 9)).cuda()
    n = t.shape[0]
    verts = torch.full((max(3 * n, 1), 3), 7.0, dtype=torch.float32, device="cuda")
    index = torch.full((max(3 * n, 1),), -1, dtype=torch.int32, device="cuda")
    nv = ctx.weld_vertices(t.data_ptr(), n, verts.data_ptr(), index.data_ptr())
    v = verts.cpu().numpy()
    assert (v[nv:] == 7.0).all()                                          # nothing beyond the count is written
    return v[:nv].view(np.uint32), index.cpu().numpy().view(np.uint32)[:3 * n]


def test_weld_dipyramid(gpu_ctx):
    vol = np.full(27, 0x00013C00, dtype=np.uint32)
    vol[13] = 0x0001BC00
    tris = gpu_mc(gpu_ctx, torch.from_numpy(vol.view(np.int32)).cuda(), (3, 3, 3), 1, 0.01)
    verts, index = gpu_weld(gpu_ctx, tris)
    assert len(verts) == 6                                                # test/marching_cubes_tests.cpp:133
    assert np.array_equal(verts[np.lexsort(verts.T[::-1])].view(np.float32),
                          DIPYRAMID_VERTICES[np.lexsort(DIPYRAMID_VERTICES.view(np.uint32).T[::-1])])
    want_v, want_i = numpy_weld(tris)
    assert np.array_equal(verts, want_v) and np.array_equal(index, want_i)
    faces = index.reshape(-1, 3)
    assert len(faces) == 8 and (faces[:, 0] != faces[:, 1]).all() and (faces[:, 1] != faces[:, 2]).all()
    # every corner got the vertex with its own bits
    assert np.array_equal(verts[index], tris.reshape(-1, 3).view(np.uint32))


@pytest.mark.parametrize("n,r", [(64, 20.3), (256, 100.7)])
def test_weld_sphere_equals_numpy(gpu_ctx, n, r):
    vol = sphere_volume(n, r)
    tris = gpu_mc(gpu_ctx, torch.from_numpy(vol.view(np.int32)).cuda(), (n, n, n), 1, 0.01)
    verts, index = gpu_weld(gpu_ctx, tris)
    want_v, want_i = numpy_weld(tris)
    assert np.array_equal(verts, want_v) and np.array_equal(index, want_i)
    assert len(verts) < len(tris) * 3 // 4                                # a closed surface: ~6 corners share a vertex


def test_weld_signed_zeros_duplicates_and_edges(gpu_ctx):
    rng = np.random.default_rng(5)
    pool = rng.standard_normal((50, 3)).astype(np.float32)
    pool[0] = (0.0, -0.0, 0.0)                                            # -0.0 and +0.0 are different bit patterns
    pool[1] = (0.0, 0.0, 0.0)
    pool[2] = (-1.5, 2.0, np.float32(1e-40))                              # negative (bit 31 set sorts last), subnormal
    tris = pool[rng.integers(0, 50, size=(1000, 3))]
    verts, index = gpu_weld(gpu_ctx, tris)
    want_v, want_i = numpy_weld(tris)
    assert np.array_equal(verts, want_v) and np.array_equal(index, want_i)
    # one triangle, three times the same corner; and no triangle at all
    verts, index = gpu_weld(gpu_ctx, np.ones((1, 3, 3), dtype=np.float32))
    assert len(verts) == 1 and index.tolist() == [0, 0, 0]
    assert gpu_ctx.weld_vertices(0, 0, 0, 0) == 0
    with pytest.raises(capi.YakHipError) as e:
        gpu_ctx.weld_vertices(0, 5, 0, 0)
    assert e.value.code == -1


def arbitrary_words(dims, seed, finite=True):
    """Any 32-bit voxel words (what uploadTSDF may bring): every half -- huge, denormal, signed zero -- and every
    weight, the 'negative' ones included; with finite=True the infinite and NaN codes lose an exponent bit."""
    rng = np.random.default_rng(seed)
    words = rng.integers(0, 2 ** 32, size=int(np.prod(dims)), dtype=np.uint64).astype(np.uint32)
    if finite:
        h = (words & 0xFFFF).astype(np.uint16)
        h = np.where((h & 0x7C00) == 0x7C00, h & 0xBBFF, h)
        words = (words & 0xFFFF0000) | h
    return words


@pytest.mark.parametrize("dims,min_weight", [((70, 19, 21), 0), ((64, 8, 40), 5), ((33, 30, 18), 20000)])
def test_mc_volume_of_arbitrary_finite_words(gpu_ctx, dims, min_weight):
    vol = arbitrary_words(dims, sum(dims) + 7)
    want = oracle.marching_cubes(vol, dims, min_weight, 0.37)
    got = gpu_mc(gpu_ctx, torch.from_numpy(vol.view(np.int32)).cuda(), dims, min_weight, 0.37)
    assert len(got) == len(want) > 50
    assert np.array_equal(canon(got), canon(want))
