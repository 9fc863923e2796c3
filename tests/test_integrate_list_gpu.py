"""The two-kernel launch of yakhip_integrate (yak_amd/csrc/integrate.hip: classify_kernel + integrate_list_kernel; the
"defer_mixed" option: 0 one kernel, 1 the launcher's choice, 2 two kernels wherever the layout allows).  It is a
schedule, not another algorithm: the voxels must equal the oracle's (reference yak/src/cuda/tsdf_volume.cu:51-110) and
the one-kernel launch's word for word -- on whole volumes and slabs, from layer 0 and from a checkpoint, with and
without the second look at a brick's parts, with the surface map, and whatever the launcher chooses from frame to
frame."""
import numpy as np
import pytest
import torch

import oracle
from yak_amd import capi, synth

pytestmark = pytest.mark.gpu


def _scene(frame, rows=synth.ROWS, cols=synth.COLS):
    depth, R, t = synth.scene_s2(frame, noise_mm=1.0)
    return oracle.compute_dists(depth, synth.INTR), R, t


def _run(ctx, dims, vs, tr, frames, z0=0, z1=None, options=(), surface_map=False):
    z1 = dims[2] if z1 is None else z1
    for k, v in options:
        ctx.set_option(k, v)
    vol = torch.zeros(dims[0] * dims[1] * (z1 - z0), dtype=torch.int32, device="cuda")
    smap = None
    desc = capi.volume_desc(vol.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT, z0, z1)
    if surface_map:
        smap = torch.zeros(capi.surface_map_bytes(desc) // 4, dtype=torch.int32, device="cuda")
        desc = capi.volume_desc(vol.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT, z0, z1, surface_map=smap.data_ptr())
    keep = []
    for dists, R, t in frames:
        d = torch.from_numpy(dists.view(np.int16)).cuda()
        keep.append(d)
        ctx.integrate(desc, d.data_ptr(), dists.shape[1] * 2, dists.shape[0], dists.shape[1], R.reshape(-1), t, synth.INTR)
        torch.cuda.synchronize()  # (one wait per frame, as a tracking loop has: the launcher's findings stay fresh)
    out = vol.cpu().numpy().view(np.uint32).copy()
    bits = smap.cpu().numpy().view(np.uint32).copy() if surface_map else None
    for k, _ in options:
        ctx.set_option(k, {"defer_mixed": 1, "refine": 1}.get(k, 0))
    return out, bits


def _oracle(dims, vs, tr, frames, z0=0, z1=None):
    z1 = dims[2] if z1 is None else z1
    want = np.zeros(dims[0] * dims[1] * dims[2], dtype=np.uint32)
    for dists, R, t in frames:
        oracle.integrate(want, dims, vs, tr, synth.MAX_WEIGHT, dists, R, t, synth.INTR)
    plane = dims[0] * dims[1]
    return want[z0 * plane:z1 * plane]


@pytest.mark.parametrize("n,frames", [(96, (0, 40, 80)), (128, (150, 151)), ((160, 96, 72), (220,))])
def test_two_kernels_give_the_oracles_voxels(gpu_ctx, n, frames):
    dims, vs, tr = synth.volume_geometry(n)
    fr = [_scene(f) for f in frames]
    want = _oracle(dims, vs, tr, fr)
    assert np.count_nonzero(want >> 16) > 0
    for opts in ((("defer_mixed", 2),), (("defer_mixed", 2), ("refine", 0)), (("defer_mixed", 2), ("mixed_waves", 2)),
                 (("defer_mixed", 2), ("z_chunk", 40)), (("defer_mixed", 0),)):
        got, _ = _run(gpu_ctx, dims, vs, tr, fr, options=opts)
        assert np.array_equal(got, want), "%s: %d voxel words differ" % (opts, np.count_nonzero(got != want))


@pytest.mark.parametrize("z0,z1", [(0, 40), (40, 100), (72, 128), (100, 128)])
def test_two_kernels_on_slabs_from_layer_0_and_from_a_checkpoint(gpu_ctx, z0, z1):
    """slabs below layer 64 replay from layer 0, deeper ones start from the closed-form column checkpoint: the table of
    the two-kernel launch begins at either"""
    dims, vs, tr = synth.volume_geometry(128)
    fr = [_scene(f) for f in (10, 95)]
    want = _oracle(dims, vs, tr, fr, z0, z1)
    got, _ = _run(gpu_ctx, dims, vs, tr, fr, z0, z1, options=(("defer_mixed", 2),))
    assert np.array_equal(got, want), "%d voxel words differ" % np.count_nonzero(got != want)


def test_two_kernels_mark_the_surface_map_like_one(gpu_ctx):
    dims, vs, tr = synth.volume_geometry(128)
    fr = [_scene(f) for f in (0, 60)]
    one, bits_one = _run(gpu_ctx, dims, vs, tr, fr, options=(("defer_mixed", 0),), surface_map=True)
    two, bits_two = _run(gpu_ctx, dims, vs, tr, fr, options=(("defer_mixed", 2),), surface_map=True)
    assert np.array_equal(one, two)
    # every brick that holds a negative voxel is marked (the map's invariant); the two launches cut the walks
    # differently, so one may mark a superset of the other's bricks -- compare against the voxels, not each other
    neg = ((two & 0x8000) != 0) & ((two & 0x7fff) != 0)
    z, y, x = np.nonzero(neg.reshape(dims[2], dims[1], dims[0]))
    bx, by = (dims[0] + 15) // 16, (dims[1] + 15) // 16
    brick = (x // 16) + bx * ((y // 16) + by * (z // 16))
    for bits in (bits_one, bits_two):
        assert np.all((bits[brick >> 5] >> (brick & 31).astype(np.uint32)) & 1), "a brick with a negative voxel is not marked"


def test_the_launchers_choice_changes_no_voxel_512(gpu_ctx):
    """512^3 (a launch large enough for the choice to exist): wall frames, then orbit frames, then wall frames again, a
    wait per frame so that the findings arrive -- against the same sequence with one kernel throughout.  The counters
    say that both launches were used."""
    dims, vs, tr = synth.volume_geometry(512)
    d0, R0, t0 = synth.scene_s0()
    wall = (oracle.compute_dists(d0, synth.INTR), R0, t0)
    fr = [wall] * 3 + [_scene(f) for f in (0, 1, 2, 3, 4, 5)] + [wall] * 3
    one, _ = _run(gpu_ctx, dims, vs, tr, fr, options=(("defer_mixed", 0),))
    before = gpu_ctx.launch_counts()
    auto, _ = _run(gpu_ctx, dims, vs, tr, fr, options=(("defer_mixed", 1),))
    after = gpu_ctx.launch_counts()
    assert np.array_equal(one, auto), "%d voxel words differ" % np.count_nonzero(one != auto)
    assert after["two_kernels"] - before["two_kernels"] >= 2 and after["one_kernel"] - before["one_kernel"] >= 6, (before, after)
    # wall frames: every voxel updated once per frame
    assert np.all((auto >> 16) >= 6)
