"""GPU parity: yakhip_extract_cloud / yakhip_extract_normals (through the C ABI) against the CPU oracle.
Bar: the same point SET, bit for bit (the reference's own order comes from a global atomic counter); normals
bit for bit, point by point."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

import oracle  # noqa: E402
from test_mc_gpu import fused, noise_volume  # noqa: E402
from test_oracle_mc import sphere_volume  # noqa: E402
from yak_amd import capi, synth  # noqa: E402

pytestmark = pytest.mark.gpu

POSE_R = np.float32([[0.36, 0.48, -0.8], [-0.8, 0.6, 0.0], [0.48, 0.64, 0.6]])
POSE_T = np.float32([0.25, -1.5, 3.0])


def canon(points):
    rows = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 4).view(np.uint32)
    return rows[np.lexsort(rows.T[::-1])]


def gpu_cloud(ctx, dev, dims, vs, R=POSE_R, t=POSE_T, z0=0, z1=None, slack=5):
    desc = capi.volume_desc(dev.data_ptr(), dims, vs, 0.04, 64, z0, z1)
    n = ctx.extract_cloud(desc, R.reshape(-1), t)
    buf = torch.full((n + slack, 4), 7.0, dtype=torch.float32, device="cuda")
    assert ctx.extract_cloud(desc, R.reshape(-1), t, buf.data_ptr(), n + slack) == n
    out = buf.cpu().numpy()
    assert (out[n:] == 7.0).all()
    return out[:n], buf


@pytest.mark.parametrize("dims", [(64, 64, 64), (130, 33, 18), (70, 19, 21), (5, 5, 70), (63, 4, 9), (64, 5, 2)])
def test_cloud_of_noise_volumes_every_edge_and_tile_boundary(gpu_ctx, dims):
    vol = noise_volume(dims, sum(dims) + 1)
    vs = (0.01, 0.013, 0.009)
    want = oracle.extract_cloud(vol, dims, vs, POSE_R, POSE_T)
    got, _ = gpu_cloud(gpu_ctx, torch.from_numpy(vol.view(np.int32)).cuda(), dims, vs)
    assert len(got) == len(want) > 30
    assert np.array_equal(canon(got), canon(want))


def test_cloud_and_normals_of_a_fused_scene(gpu_ctx):
    vol, dims = fused((96, 80, 112), [0, 20, 40, 60, 80, 100])
    vs = tuple(3.0 / 96 for _ in range(3))
    Rinv = np.ascontiguousarray(POSE_R.T)
    want = oracle.extract_cloud(vol, dims, vs, POSE_R, POSE_T)
    dev = torch.from_numpy(vol.view(np.int32)).cuda()
    got, buf = gpu_cloud(gpu_ctx, dev, dims, vs)
    assert len(got) == len(want) > 3000 and np.array_equal(canon(got), canon(want))
    # normals for the GPU's own point order
    nrm = torch.full((len(got), 4), 7.0, dtype=torch.float32, device="cuda")
    desc = capi.volume_desc(dev.data_ptr(), dims, vs, 0.04, 64)
    gpu_ctx.extract_normals(desc, buf.data_ptr(), len(got), POSE_R.reshape(-1), POSE_T, Rinv.reshape(-1), nrm.data_ptr())
    torch.cuda.synchronize()
    want_n = oracle.extract_normals(vol, dims, vs, got, POSE_R, POSE_T, Rinv)
    got_n = nrm.cpu().numpy()
    nan = np.isnan(want_n[:, 0])
    assert nan.sum() < len(got) // 2 and np.array_equal(np.isnan(got_n[:, 0]), nan)
    assert np.array_equal(got_n[~nan].view(np.uint32), want_n[~nan].view(np.uint32))
    assert np.isnan(got_n[nan, :3]).all() and (got_n[:, 3] == 0).all()


@pytest.mark.parametrize("dims", [(40, 33, 18), (9, 70, 12)])
def test_normals_of_noise_volumes_including_the_border_band(gpu_ctx, dims):
    """Points of a noise volume lie everywhere, also within two voxels of the faces, where the reference
    writes NaN (tsdf_volume.cu:700-707)."""
    vol = noise_volume(dims, sum(dims) + 3)
    vs = (0.01, 0.013, 0.009)
    Rinv = np.ascontiguousarray(POSE_R.T)
    dev = torch.from_numpy(vol.view(np.int32)).cuda()
    got, buf = gpu_cloud(gpu_ctx, dev, dims, vs)
    nrm = torch.full((len(got), 4), 7.0, dtype=torch.float32, device="cuda")
    desc = capi.volume_desc(dev.data_ptr(), dims, vs, 0.04, 64)
    gpu_ctx.extract_normals(desc, buf.data_ptr(), len(got), POSE_R.reshape(-1), POSE_T, Rinv.reshape(-1), nrm.data_ptr(),
                            gradient_delta_factor=0.75)
    torch.cuda.synchronize()
    want_n = oracle.extract_normals(vol, dims, vs, got, POSE_R, POSE_T, Rinv, gradient_delta_factor=0.75)
    got_n = nrm.cpu().numpy()
    nan = np.isnan(want_n[:, 0])
    assert 0 < nan.sum() < len(got) and np.array_equal(np.isnan(got_n[:, 0]), nan)
    assert np.array_equal(got_n[~nan].view(np.uint32), want_n[~nan].view(np.uint32))


def test_cloud_slabs_capacity_and_degenerate_volumes(gpu_ctx):
    vol = sphere_volume(48, 15.2)
    dims, vs = (48, 48, 48), (0.02, 0.02, 0.02)
    dev = torch.from_numpy(vol.view(np.int32)).cuda()
    whole, _ = gpu_cloud(gpu_ctx, dev, dims, vs)
    assert np.array_equal(canon(whole), canon(oracle.extract_cloud(vol, dims, vs, POSE_R, POSE_T)))
    plane = 48 * 48
    parts = [gpu_cloud(gpu_ctx, dev[z0 * plane:], dims, vs, z0=z0, z1=z1)[0] for z0, z1 in [(0, 17), (16, 30), (29, 48)]]
    assert sum(len(p) for p in parts) == len(whole) and np.array_equal(canon(np.concatenate(parts)), canon(whole))
    # a buffer that is too small receives the first `capacity` points of the library's order, nothing beyond
    desc = capi.volume_desc(dev.data_ptr(), dims, vs, 0.04, 64)
    n = len(whole)
    small = torch.full((n, 4), 7.0, dtype=torch.float32, device="cuda")
    assert gpu_ctx.extract_cloud(desc, POSE_R.reshape(-1), POSE_T, small.data_ptr(), 100) == n
    out = small.cpu().numpy()
    assert np.array_equal(out[:100], whole[:100]) and (out[100:] == 7.0).all()
    one = capi.volume_desc(dev.data_ptr(), dims, vs, 0.04, 64, 5, 6)
    assert gpu_ctx.extract_cloud(one, POSE_R.reshape(-1), POSE_T) == 0          # a single layer has no z + 1
    with pytest.raises(capi.YakHipError):
        gpu_ctx.extract_cloud(desc, POSE_R.reshape(-1), POSE_T, small.data_ptr() + 4, n)


def test_cloud_512_volume_fused_on_gpu(gpu_ctx):
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
    got, _ = gpu_cloud(gpu_ctx, vol, dims, vs)
    want = oracle.extract_cloud(vol.cpu().numpy().view(np.uint32), dims, vs, POSE_R, POSE_T)
    assert len(got) == len(want) > 300000
    assert np.array_equal(canon(got), canon(want))


@pytest.mark.parametrize("dims", [(70, 19, 21), (64, 8, 40)])
def test_cloud_of_a_volume_of_arbitrary_finite_words(gpu_ctx, dims):
    from test_mc_gpu import arbitrary_words

    vol = arbitrary_words(dims, sum(dims) + 3)
    vs = (0.01, 0.013, 0.009)
    want = oracle.extract_cloud(vol, dims, vs, POSE_R, POSE_T)
    got, _ = gpu_cloud(gpu_ctx, torch.from_numpy(vol.view(np.int32)).cuda(), dims, vs)
    assert len(got) == len(want) > 30
    assert np.array_equal(canon(got), canon(want))


def test_normals_at_points_that_are_not_a_cloud(gpu_ctx):
    """fetchNormals takes any points (tsdf_volume.cu:689-722): far outside the volume, on its faces, NaN and infinite
    coordinates, over a volume of arbitrary finite words.  NaN where the oracle has NaN, its bits elsewhere."""
    from test_mc_gpu import arbitrary_words

    rng = np.random.default_rng(23)
    dims, vs = (40, 33, 18), (0.01, 0.013, 0.009)
    vol = arbitrary_words(dims, 41)
    Rinv = np.ascontiguousarray(POSE_R.T)
    ext = np.float32([dims[0] * vs[0], dims[1] * vs[1], dims[2] * vs[2]])
    local = rng.uniform(-0.3, 1.3, size=(5000, 3)).astype(np.float32) * ext   # volume frame, a third of them outside
    local[:200] = np.round(local[:200] / np.float32(vs)) * np.float32(vs)      # on voxel centres
    local[200:260, 0] = 0.0
    local[260:320, 2] = ext[2]
    world = (local @ POSE_R.T + POSE_T).astype(np.float32)                    # pose: volume -> world (as extract_cloud maps)
    world[320:330] = np.nan
    world[330:335, 1] = np.inf
    world[335:340] = np.float32(1e30)
    pts = np.zeros((len(world), 4), dtype=np.float32)
    pts[:, :3] = world
    dev = torch.from_numpy(vol.view(np.int32)).cuda()
    dpts = torch.from_numpy(pts).cuda()
    nrm = torch.full((len(pts), 4), 7.0, dtype=torch.float32, device="cuda")
    desc = capi.volume_desc(dev.data_ptr(), dims, vs, 0.04, 64)
    gpu_ctx.extract_normals(desc, dpts.data_ptr(), len(pts), POSE_R.reshape(-1), POSE_T, Rinv.reshape(-1), nrm.data_ptr())
    torch.cuda.synchronize()
    want = oracle.extract_normals(vol, dims, vs, pts, POSE_R, POSE_T, Rinv)
    got = nrm.cpu().numpy()
    nan = np.isnan(want)
    assert 100 < np.isnan(want[:, 0]).sum() < len(pts) - 100
    assert np.array_equal(np.isnan(got), nan)
    assert np.array_equal(got[~nan].view(np.uint32), want[~nan].view(np.uint32))
