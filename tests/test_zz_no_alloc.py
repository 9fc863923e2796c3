"""yakhip_integrate_static: the allocation-free entry point (no counterpart in the reference; for callers that
cannot allocate: HIP graph capture, serving loops).  It must write the same voxels, bit for bit, as
yakhip_integrate -- the existing entry point whose context-owned scratch (band thresholds, coarse minima,
column checkpoints) it replaces by the library's file-scope __device__ array.

The array is shared by every call on the GPU and keeps the last call's contents, so the test makes two calls
in a row, the second with a smaller image and a smaller volume (stale words of the first call lie behind the
second call's data), and a deep slab (checkpoint part of the array).  Collected last (test_zz_*): the
round's end runs pytest -x.
"""
import numpy as np
import pytest
import torch

import oracle
from yak_amd import capi, synth

pytestmark = pytest.mark.gpu


def _frame(rows, cols, seed):
    """A ragged synthetic ray-distance image: a tilted plane with holes, as half bits."""
    rng = np.random.default_rng(seed)
    v, u = np.mgrid[0:rows, 0:cols].astype(np.float32)
    depth = (2200.0 + 1.3 * u - 0.7 * v + rng.normal(0.0, 2.0, (rows, cols))).astype(np.float32)
    depth[rng.random((rows, cols)) < 0.05] = 0.0
    depth = np.clip(depth, 0, 65535).astype(np.uint16)
    intr = (0.9 * cols, 0.9 * cols, cols / 2 - 0.5, rows / 2 - 0.5)
    return oracle.compute_dists(depth, intr), intr


def _pose(dims, vs):
    ex, ey = dims[0] * vs[0], dims[1] * vs[1]
    R = np.eye(3, dtype=np.float32)
    t = np.array([-ex / 2, -ey / 2, 1.2], dtype=np.float32)
    return R, t


def _run(ctx, entry, dims, vs, tr, dists, intr, R, t, z0, z1, frames):
    n = dims[0] * dims[1] * (z1 - z0)
    vol = torch.zeros(n, dtype=torch.int32, device="cuda")
    rows, cols = dists.shape
    dd = torch.from_numpy(dists.view(np.int16)).cuda()
    desc = capi.volume_desc(vol.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT, z0, z1)
    for _ in range(frames):
        getattr(ctx, entry)(desc, dd.data_ptr(), cols * 2, rows, cols, R.reshape(-1), t, intr)
    torch.cuda.synchronize()
    return vol.cpu().numpy().view(np.uint32).copy()


CASES = [
    # (dims, z0, z1, image rows x cols) -- a large call first, then smaller ones
    ((128, 96, 160), 0, 160, (480, 640)),
    ((64, 48, 40), 0, 40, (120, 160)),     # smaller image, smaller volume
    ((64, 64, 256), 128, 200, (240, 320)),  # deep slab: column checkpoints from the array's second part
    ((32, 32, 96), 70, 96, (60, 80)),       # smaller again, still a deep slab
    ((30, 22, 19), 0, 19, (37, 53)),        # 8-byte path, odd image
]


def test_static_entry_equals_allocating_entry_on_consecutive_calls(gpu_ctx):
    for k, (dims, z0, z1, (rows, cols)) in enumerate(CASES):
        voxel = 2.2 / (z0 + z1)  # the middle layer of the slab lies 2.3 m from the camera, where the surface is
        vs = (voxel, voxel, voxel)
        tr = max(0.06, 2.1 * voxel)
        dists, intr = _frame(rows, cols, 100 + k)
        R, t = _pose(dims, vs)
        want = _run(gpu_ctx, "integrate", dims, vs, tr, dists, intr, R, t, z0, z1, frames=2)
        got = _run(gpu_ctx, "integrate_static", dims, vs, tr, dists, intr, R, t, z0, z1, frames=2)
        assert np.count_nonzero(want >> 16) > 0, "case %d updates nothing" % k
        assert np.array_equal(got >> 16, want >> 16), "case %d: weights differ" % k
        assert np.array_equal(got, want), "case %d: %d voxel words differ" % (k, np.count_nonzero(got != want))


def test_static_entry_against_the_oracle(gpu_ctx):
    dims, vs, tr = synth.volume_geometry(64)
    depth, R, t = synth.scene_s1()
    dists = oracle.compute_dists(depth, synth.INTR)
    want = np.zeros(64 ** 3, dtype=np.uint32)
    oracle.integrate(want, dims, vs, tr, synth.MAX_WEIGHT, dists, R, t, synth.INTR)
    got = _run(gpu_ctx, "integrate_static", dims, vs, tr, dists, synth.INTR, R, t, 0, 64, frames=1)
    assert np.array_equal(got, want)


def test_static_entry_rejects_an_image_beyond_its_scratch(gpu_ctx):
    rows, cols = 1025, 1024  # one row more than YAKHIP_STATIC_MAX_PIXELS = 2^20
    dd = torch.zeros((rows, cols), dtype=torch.int16, device="cuda")
    vol = torch.zeros(16 ** 3, dtype=torch.int32, device="cuda")
    desc = capi.volume_desc(vol.data_ptr(), (16, 16, 16), (0.1, 0.1, 0.1), 0.3, 64)
    with pytest.raises(capi.YakHipError) as e:
        gpu_ctx.integrate_static(desc, dd.data_ptr(), cols * 2, rows, cols, np.eye(3).reshape(-1), np.zeros(3),
                                 (500.0, 500.0, 512.0, 512.0))
    assert e.value.code == -4 and "static scratch" in str(e.value)
    # the allocating entry point takes the same frame
    gpu_ctx.integrate(desc, dd.data_ptr(), cols * 2, rows, cols, np.eye(3).reshape(-1), np.zeros(3),
                      (500.0, 500.0, 512.0, 512.0))
    torch.cuda.synchronize()


def test_static_entry_on_arbitrary_voxel_words_and_an_impossible_pose(gpu_ctx):
    """The same words as yakhip_integrate where the volume did not come from integrate (any halves, any weights) and
    where the culling tests switch themselves off (NaN in the pose, a negative focal length)."""
    rng = np.random.default_rng(3)
    dims, z0, z1, rows, cols = (64, 48, 40), 0, 40, 120, 160
    voxel = 2.2 / (z0 + z1)
    vs, tr = (voxel, voxel, voxel), max(0.06, 2.1 * voxel)
    dists, intr = _frame(rows, cols, 9)
    R, t = _pose(dims, vs)
    start = rng.integers(0, 2 ** 32, size=dims[0] * dims[1] * (z1 - z0), dtype=np.uint64).astype(np.uint32)
    h = (start & 0xFFFF).astype(np.uint16)
    start = (start & 0xFFFF0000) | np.where((h & 0x7C00) == 0x7C00, h & 0xBBFF, h)
    dd = torch.from_numpy(dists.view(np.int16)).cuda()
    poses = [(R, t, intr), (R, np.array([t[0], np.nan, t[2]], dtype=np.float32), intr), (R, t, (-intr[0], intr[1], intr[2], intr[3]))]
    out = {}
    for entry in ("integrate", "integrate_static"):
        vol = torch.from_numpy(start.view(np.int32)).cuda()
        desc = capi.volume_desc(vol.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT, z0, z1)
        for R2, t2, i2 in poses:
            getattr(gpu_ctx, entry)(desc, dd.data_ptr(), cols * 2, rows, cols, R2.reshape(-1), t2, i2)
        torch.cuda.synchronize()
        out[entry] = vol.cpu().numpy().view(np.uint32).copy()
    assert np.count_nonzero(out["integrate"] != start) > 1000
    assert np.array_equal(out["integrate_static"], out["integrate"])
