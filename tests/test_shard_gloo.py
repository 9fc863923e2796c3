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
