"""GPU parity: yakhip_integrate_batch (K frames folded into the volume in one sweep) against K
consecutive integrate calls -- of the CPU oracle and of the single-frame HIP kernel.  Bar: bit-exact
("bit-identical to K sequential integrate calls")."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

import oracle  # noqa: E402
from yak_amd import capi, synth  # noqa: E402

pytestmark = pytest.mark.gpu


def frames_s2(indices, noise=0.0):
    out = []
    for f in indices:
        depth, R, t = synth.scene_s2(f, noise_mm=noise)
        out.append((oracle.compute_dists(depth, synth.INTR), R, t, synth.INTR))
    return out


def run_batch(ctx, start, dims, vs, tr, max_weight, frames, z0=0, z1=None, misalign=0):
    z1 = dims[2] if z1 is None else z1
    n = dims[0] * dims[1] * (z1 - z0)
    store = torch.zeros(n + 8, dtype=torch.int32, device="cuda")
    vol = store[misalign:misalign + n]
    vol.copy_(torch.from_numpy(start.view(np.int32)).cuda())
    desc = capi.volume_desc(vol.data_ptr(), dims, vs, tr, max_weight, z0, z1)
    keep, args = [], []
    for dists, R, t, intr in frames:
        d = torch.from_numpy(dists.view(np.int16)).cuda()
        keep.append(d)
        args.append((d.data_ptr(), dists.shape[1] * 2, dists.shape[0], dists.shape[1], np.asarray(R).reshape(-1), t, intr))
    ctx.integrate_batch(desc, args)
    torch.cuda.synchronize()
    return vol.cpu().numpy().view(np.uint32).copy()


def run_oracle(start, dims, vs, tr, max_weight, frames, z0=0, z1=None):
    vol = start.copy()
    for dists, R, t, intr in frames:
        oracle.integrate(vol, dims, vs, tr, max_weight, dists, R, t, intr, z0=z0, z1=z1, nthreads=8)
    return vol


@pytest.mark.parametrize("k", [1, 2, 3, 4, 6, 9])
def test_batch_equals_sequential_oracle(gpu_ctx, k):
    dims, vs, tr = synth.volume_geometry(64)
    frames = frames_s2([0, 75, 150, 225, 30, 105, 180, 255, 12][:k], noise=1.0)
    start = np.zeros(64 ** 3, dtype=np.uint32)
    want = run_oracle(start, dims, vs, tr, synth.MAX_WEIGHT, frames)
    got = run_batch(gpu_ctx, start, dims, vs, tr, synth.MAX_WEIGHT, frames)
    assert np.count_nonzero(want) > 10000
    assert np.array_equal(got >> 16, want >> 16), "weights differ"
    assert np.array_equal(got, want), "tsdf differs"


def test_batch_mixed_scenes_second_pass_and_saturation(gpu_ctx):
    dims, vs, tr = synth.volume_geometry((128, 96, 112))
    d0, R0, t0 = synth.scene_s0()
    d1, R1, t1 = synth.scene_s1()
    frames = [(oracle.compute_dists(d0, synth.INTR), R0, t0, synth.INTR),
              (oracle.compute_dists(d1, synth.INTR), R1, t1, synth.INTR)] + frames_s2([40, 200])
    start = np.zeros(128 * 96 * 112, dtype=np.uint32)
    want = run_oracle(start, dims, vs, tr, 5, frames)
    got = run_batch(gpu_ctx, start, dims, vs, tr, 5, frames)
    assert np.array_equal(got, want)
    # again on top of the result: running averages of non-trivial values, weights saturating at 5
    want2 = run_oracle(want, dims, vs, tr, 5, frames)
    got2 = run_batch(gpu_ctx, got, dims, vs, tr, 5, frames)
    assert (want2 >> 16).max() == 5 and np.array_equal(got2, want2)


def test_batch_slabs_and_group_test_switch(gpu_ctx):
    dims, vs, tr = synth.volume_geometry(64)
    frames = frames_s2([10, 100, 190, 280])
    plane = 64 * 64
    whole = run_oracle(np.zeros(64 ** 3, dtype=np.uint32), dims, vs, tr, synth.MAX_WEIGHT, frames)
    for z0, z1 in [(0, 20), (20, 47), (47, 64)]:
        got = run_batch(gpu_ctx, np.zeros(plane * (z1 - z0), dtype=np.uint32), dims, vs, tr, synth.MAX_WEIGHT, frames,
                        z0=z0, z1=z1)
        assert np.array_equal(got, whole[z0 * plane:z1 * plane])
    gpu_ctx.set_option("group_test", 0)
    try:
        got = run_batch(gpu_ctx, np.zeros(64 ** 3, dtype=np.uint32), dims, vs, tr, synth.MAX_WEIGHT, frames)
    finally:
        gpu_ctx.set_option("group_test", 1)
    assert np.array_equal(got, whole)
    gpu_ctx.set_option("z_chunk", 5)
    try:
        got = run_batch(gpu_ctx, np.zeros(64 ** 3, dtype=np.uint32), dims, vs, tr, synth.MAX_WEIGHT, frames)
    finally:
        gpu_ctx.set_option("z_chunk", 0)
    assert np.array_equal(got, whole)


def test_batch_deep_slab_uses_one_checkpoint_per_frame(gpu_ctx):
    dims = (64, 48, 900)
    vs, tr = (3.0 / 64, 3.0 / 48, 3.0 / 900), 0.04
    frames = frames_s2([20, 110, 200, 290, 65])
    z0, z1 = 700, 760
    n = dims[0] * dims[1] * (z1 - z0)
    want = np.zeros(n, dtype=np.uint32)
    want = run_oracle(want, dims, vs, tr, synth.MAX_WEIGHT, frames, z0=z0, z1=z1)
    assert (want != 0).sum() > 1000
    got = run_batch(gpu_ctx, np.zeros(n, dtype=np.uint32), dims, vs, tr, synth.MAX_WEIGHT, frames, z0=z0, z1=z1)
    assert np.array_equal(got, want)


def test_batch_narrow_layouts_fall_back_frame_by_frame(gpu_ctx):
    frames = frames_s2([5, 95, 185])
    # X not a multiple of 4, and a 4-byte-aligned (not 16) base: the single-frame kernels take over
    dims, vs, tr = synth.volume_geometry((50, 44, 40))
    start = np.zeros(50 * 44 * 40, dtype=np.uint32)
    want = run_oracle(start, dims, vs, tr, synth.MAX_WEIGHT, frames)
    assert np.array_equal(run_batch(gpu_ctx, start, dims, vs, tr, synth.MAX_WEIGHT, frames), want)
    dims, vs, tr = synth.volume_geometry(48)
    start = np.zeros(48 ** 3, dtype=np.uint32)
    want = run_oracle(start, dims, vs, tr, synth.MAX_WEIGHT, frames)
    assert np.array_equal(run_batch(gpu_ctx, start, dims, vs, tr, synth.MAX_WEIGHT, frames, misalign=1), want)


def test_batch_512_equals_single_frame_kernel(gpu_ctx):
    """BASELINE size: four orbit cameras 90 degrees apart (config C5's rig) into 512^3.
    This is also the GPU-side regression of the 16-byte store-data hazard: it was at this
    size, in the batched kernel, that a v_cndmask scheduled into a store's data register reached memory instead of
    the voxel; tools/check_store_hazard.py guards the same thing at build time."""
    n = 512
    dims, vs, tr = synth.volume_geometry(n)
    frames = frames_s2([0, 75, 150, 225])
    devs = [torch.from_numpy(f[0].view(np.int16)).cuda() for f in frames]
    a = torch.zeros(n ** 3, dtype=torch.int32, device="cuda")
    b = torch.zeros(n ** 3, dtype=torch.int32, device="cuda")
    da = capi.volume_desc(a.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT)
    db = capi.volume_desc(b.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT)
    args = [(d.data_ptr(), synth.COLS * 2, synth.ROWS, synth.COLS, f[1].reshape(-1), f[2], f[3]) for d, f in zip(devs, frames)]
    for rep in range(2):
        gpu_ctx.integrate_batch(da, args)
        for ptr, pitch, rows, cols, R, t, intr in args:
            gpu_ctx.integrate(db, ptr, pitch, rows, cols, R, t, intr)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    assert int((a >> 16).max().item()) == 8 and int((a != 0).sum().item()) > 10_000_000


def test_batch_rejects_bad_arguments(gpu_ctx):
    dims, vs, tr = synth.volume_geometry(16)
    vol = torch.zeros(16 ** 3, dtype=torch.int32, device="cuda")
    desc = capi.volume_desc(vol.data_ptr(), dims, vs, tr, 64)
    img = torch.zeros((480, 640), dtype=torch.int16, device="cuda")
    eye, t = np.eye(3).reshape(-1), np.zeros(3)
    with pytest.raises(capi.YakHipError) as e:
        gpu_ctx.integrate_batch(desc, [])
    assert e.value.code == -1
    with pytest.raises(capi.YakHipError) as e:   # image sizes differ
        gpu_ctx.integrate_batch(desc, [(img.data_ptr(), 1280, 480, 640, eye, t, synth.INTR),
                                       (img.data_ptr(), 1280, 240, 640, eye, t, synth.INTR)])
    assert e.value.code == -1
    with pytest.raises(capi.YakHipError) as e:   # null image
        gpu_ctx.integrate_batch(desc, [(0, 1280, 480, 640, eye, t, synth.INTR)])
    assert e.value.code == -1


@pytest.mark.parametrize("bad_at", [0, 2, 3])
def test_batch_with_a_frame_whose_pose_no_camera_has(gpu_ctx, bad_at):
    """One frame of the four has a non-finite, a mirrored or a negative-focal-length pose: the culling tests are per
    workgroup, so they are off for the whole sweep; every frame must still be folded as the oracle folds it, in order,
    over a volume of arbitrary finite words."""
    rng = np.random.default_rng(5 + bad_at)
    dims, vs, tr = synth.volume_geometry((64, 48, 40))
    frames = frames_s2([0, 75, 150, 225], noise=1.0)
    dists, R, t, intr = frames[bad_at]
    R = np.asarray(R, dtype=np.float32).reshape(3, 3)
    frames[bad_at] = [(dists, R, np.array([t[0], np.nan, t[2]], dtype=np.float32), intr),
                      (dists, R * np.array([[-1.0], [1.0], [1.0]], dtype=np.float32), t * np.array([-1, 1, 1], dtype=np.float32), intr),
                      (dists, R, t, (-intr[0], intr[1], intr[2], intr[3]))][bad_at % 3]
    start = rng.integers(0, 2 ** 32, size=64 * 48 * 40, dtype=np.uint64).astype(np.uint32)
    h = (start & 0xFFFF).astype(np.uint16)
    start = (start & 0xFFFF0000) | np.where((h & 0x7C00) == 0x7C00, h & 0xBBFF, h)
    want = run_oracle(start, dims, vs, tr, synth.MAX_WEIGHT, frames)
    got = run_batch(gpu_ctx, start, dims, vs, tr, synth.MAX_WEIGHT, frames)
    assert np.count_nonzero(want != start) > 10000
    assert np.array_equal(got, want), "%d voxel words differ" % np.count_nonzero(got != want)


@pytest.mark.parametrize("rows,cols", [(1, 1), (2, 3), (1080, 1920)])
def test_batch_image_sizes_from_one_pixel_to_full_hd(gpu_ctx, rows, cols):
    rng = np.random.default_rng(rows + cols)
    intr = (0.8 * max(cols, 8), 0.8 * max(cols, 8), cols / 2 - 0.5, rows / 2 - 0.5)
    dims, vs, tr = synth.volume_geometry((56, 44, 48))
    ex, ey = dims[0] * vs[0], dims[1] * vs[1]
    frames = []
    for k in range(3):
        depth = (2000.0 + 150.0 * k + rng.normal(0, 3, (rows, cols))).clip(0, 65535).astype(np.uint16)
        R = np.eye(3, dtype=np.float32)
        t = np.array([-ex / 2 + 0.05 * k, -ey / 2, 0.8], dtype=np.float32)
        frames.append((oracle.compute_dists(depth, intr), R, t, intr))
    start = np.zeros(56 * 44 * 48, dtype=np.uint32)
    want = run_oracle(start, dims, vs, tr, synth.MAX_WEIGHT, frames)
    got = run_batch(gpu_ctx, start, dims, vs, tr, synth.MAX_WEIGHT, frames)
    assert np.count_nonzero(want) > 0
    assert np.array_equal(got, want), "%d voxel words differ" % np.count_nonzero(got != want)
