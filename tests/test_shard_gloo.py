"""N > 1 host path on CPU: two gloo ranks, z-slab partition, timing reduction, slab gather.
The slabs' contents come from the oracle (the checker) so the gathered volume can be compared with a
whole-volume run -- the same property the GPU test checks for the HIP kernel on one device."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_slab_ranges_tile_the_volume():
    from yak_amd.shard import slab_range

    for layers in (1, 7, 512, 1000, 1024):
        for world in (1, 2, 3, 4, 8):
            edges = [slab_range(layers, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == layers
            assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        slab_range(8, 2, 2)


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    import oracle
    from yak_amd import synth
    from yak_amd.shard import gather_volume, reduce_timing, slab_range

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dims, vs, tr = synth.volume_geometry((24, 20, 27))
        depth, R, t = synth.scene_s2(31)
        dists = oracle.compute_dists(depth, synth.INTR)
        z0, z1 = slab_range(dims[2], world, rank)
        slab = np.zeros(dims[0] * dims[1] * (z1 - z0), dtype=np.uint32)
        _, updated = oracle.integrate(slab, dims, vs, tr, 64, dists, R, t, synth.INTR, z0=z0, z1=z1)
        elapsed, total = reduce_timing(dist, 1.0 + rank, updated)
        whole = gather_volume(dist, torch.from_numpy(slab.view(np.int32)), dims, dst=0)
        if rank == 0:
            want = np.zeros(dims[0] * dims[1] * dims[2], dtype=np.uint32)
            _, upd_all = oracle.integrate(want, dims, vs, tr, 64, dists, R, t, synth.INTR)
            ok = np.array_equal(whole.numpy().view(np.uint32), want) and total == upd_all and elapsed == float(world)
            with open(out_path, "w") as f:
                f.write("ok" if ok else "mismatch total=%s upd=%s elapsed=%s" % (total, upd_all, elapsed))
        else:
            assert whole is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_two_rank_slabs_gather_to_the_whole_volume(tmp_path, world):
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert open(out).read() == "ok"


def _tracking_worker(rank, world, port, out_path):
    """The two exchange steps of the tracking side over gloo: every rank ray-casts its own z-slab (the
    oracle's slab form stands in for the HIP kernel) and the slabs are composited with two all-reduces;
    every rank accumulates the ICP sums of its band of rows and the 33 values are all-reduced."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    import oracle
    from test_oracle_raycast import _cam2vol, fused_orbit_volume
    from yak_amd import synth
    from yak_amd.shard import allreduce_icp_sums, composite_raycast, row_range, slab_range

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        vol, dims, vs, tr = fused_orbit_volume(40, [0, 75, 150, 225])
        rows, cols, intr = 48, 64, (58.0, 58.0, 31.5, 23.5)
        _, R_v2c, t_v2c = synth.scene_s2(40)
        R, t = _cam2vol(R_v2c, t_v2c)
        Rinv = np.ascontiguousarray(R.T)
        z0, z1 = slab_range(dims[2], world, rank)
        msgs = []
        for want_depth in (False, True):
            img, nrm, keys = oracle.raycast_slab(vol, dims, vs, tr, R, t, Rinv, intr, rows, cols, z0, z1,
                                                 want_depth=want_depth)
            first = torch.from_numpy(img.view(np.int16)) if want_depth else torch.from_numpy(img)
            (c_img, c_nrm), _, flagged = composite_raycast(dist, torch.from_numpy(keys), [first, torch.from_numpy(nrm)],
                                                           [0 if want_depth else float("nan"), float("nan")])
            w_img, w_nrm, _, hits = oracle.raycast(vol, dims, vs, tr, R, t, Rinv, intr, rows, cols, want_depth=want_depth)
            same = (np.array_equal(c_img.numpy().view(np.uint16 if want_depth else np.uint32),
                                   w_img.view(np.uint16 if want_depth else np.uint32))
                    and np.array_equal(c_nrm.numpy().view(np.uint32), w_nrm.view(np.uint32)))
            if not (same and flagged == 0 and hits > 300):
                msgs.append("raycast depth=%s same=%s flagged=%d hits=%d" % (want_depth, same, flagged, hits))

        # second step of the composite on made-up data: rank 0 wins pixels 3 and 5 with odd keys (deferred
        # normals); the "slab" that holds pixel 3's point is the last rank, pixel 5's is rank 0
        keys_s = torch.full((2, 4), 0x7FFFFFFF, dtype=torch.int32)
        pay = torch.zeros((2, 4, 4), dtype=torch.float32)
        if rank == 0:
            keys_s.view(-1)[3], keys_s.view(-1)[5], keys_s.view(-1)[6] = 11, 21, 8
            pay.view(-1, 4)[3], pay.view(-1, 4)[5], pay.view(-1, 4)[6] = 3.0, 5.0, 6.0
        else:
            keys_s.view(-1)[3], keys_s.view(-1)[1] = 40 + 2 * rank, 2 * rank          # later pairs; pixel 1: rank 1 wins

        def fake_resolve(k, images):
            for pix, owner in ((3, world - 1), (5, 0)):
                if rank == owner:
                    assert k.view(-1)[pix] & 1 and images[0].view(-1, 4)[pix, 0] == float(pix)
                    images[0].view(-1, 4)[pix] = 100.0 + pix
                    k.view(-1)[pix] -= 1

        (c_pay,), c_keys, left = composite_raycast(dist, keys_s, [pay], [float("nan")], resolve=fake_resolve)
        want_k = [0x7FFFFFFF, 2 * 1, 0x7FFFFFFF, 10, 0x7FFFFFFF, 20, 8, 0x7FFFFFFF]
        want_v = [float("nan"), 0.0, float("nan"), 103.0, float("nan"), 105.0, 6.0, float("nan")]
        got_v = c_pay.view(-1, 4)[:, 0].numpy()
        if not (left == 0 and c_keys.view(-1).tolist() == want_k and np.array_equal(got_v, np.float32(want_v), equal_nan=True)):
            msgs.append("second step: left=%d keys=%s values=%s" % (left, c_keys.view(-1).tolist(), got_v))

        # ICP: the frame's own maps against the model maps ray-cast above, a slightly wrong pose
        depth, _, _ = synth.scene_s2(40)
        small = np.ascontiguousarray(depth[::10, ::10])
        vcurr, ncurr = oracle.compute_points_normals(small, intr)
        vprev, nprev, _, _ = oracle.raycast(vol, dims, vs, tr, R, t, Rinv, intr, rows, cols)
        Rg = np.eye(3, dtype=np.float32)
        tg = np.float32([0.004, -0.003, 0.002])
        r0, r1 = row_range(rows, world, rank)
        band_v = np.full_like(vcurr, np.nan)                       # rows outside the band: "no source point"
        band_v[r0:r1] = vcurr[r0:r1]
        sums, counts = oracle.icp_accumulate(Rg, tg, intr, 0, 0.1, 0.6, band_v, ncurr, vprev, nprev)
        mine = torch.from_numpy(np.concatenate([sums, counts.astype(np.float64)]).astype(np.float32))
        mine[28] -= float(cols * (rows - (r1 - r0)))              # the masked rows are not this rank's rejections
        total = allreduce_icp_sums(dist, mine).numpy()
        w_sums, w_counts = oracle.icp_accumulate(Rg, tg, intr, 0, 0.1, 0.6, vcurr, ncurr, vprev, nprev)
        scale = np.abs(w_sums).max()
        if not (w_counts[0] > 200 and np.array_equal(total[27:], w_counts.astype(np.float32))
                and np.abs(total[:27] - w_sums).max() <= 1e-4 * scale):
            msgs.append("icp counts %s vs %s, err %g" % (total[27:], w_counts, np.abs(total[:27] - w_sums).max() / scale))
        if rank == 0:
            with open(out_path, "w") as f:
                f.write("ok" if not msgs else "; ".join(msgs))
        else:
            assert not msgs, msgs
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_slab_raycast_composite_and_icp_sums_over_ranks(tmp_path, world):
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "result.txt")
    mp.spawn(_tracking_worker, args=(world, port, out), nprocs=world, join=True)
    assert open(out).read() == "ok"
