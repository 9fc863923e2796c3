"""The KinFu loop over a z-slab sharded volume (yak_amd/kinfu_sharded.py; BASELINE config 4 in small):
two and three processes share the one GPU of the test box, each holding its own slab + halo, talking
over gloo.  Checked against the same loop run by a single rank on the whole volume:
  * supplied poses: own slabs concatenate to the single-rank volume and the composited model maps equal
    its ray cast, bit for bit, on every frame;
  * tracking from depth alone: the all-reduced ICP sums differ from the one-launch sums by rounding, so
    poses agree closely (not bitwise) and the track stays on the orbit."""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, FRAMES = 96, 7


def c2v(R_v2c, t_v2c):
    R = R_v2c.astype(np.float64).T
    return R, -R @ t_v2c.astype(np.float64)


def run_loop(dist, track, out, blank_at=None):
    from yak_amd import capi, synth
    from yak_amd.kinfu_sharded import ShardedKinFu

    ctx = capi.Context(0)
    dims, vs, tr = synth.volume_geometry(N)
    kf = ShardedKinFu(ctx, dist, dims, vs, tr, synth.MAX_WEIGHT, synth.INTR, synth.ROWS, synth.COLS)
    maps, poses = [], []
    for f in range(FRAMES):
        depth, R_v2c, t_v2c = synth.scene_s2(3 * f)
        if f == blank_at:
            depth = np.zeros_like(depth)  # a frame without a single return: ICP has nothing to hold on to
        d = torch.from_numpy(depth.view(np.int16)).cuda()
        true_pose = c2v(R_v2c, t_v2c)
        fused = kf(d, pose=None if track else true_pose, first_pose=true_pose)
        if blank_at is None:
            assert fused == (f > 0)
        out.setdefault("fused", []).append(bool(fused))
        out.setdefault("frame_counter", []).append(int(kf.frame))
        poses.append((kf.pose_R.copy(), kf.pose_t.copy(), true_pose[1]))
        maps.append((kf.prv_p[0].cpu().numpy().copy(), kf.prv_n[0].cpu().numpy().copy()))
    out.update(volume=kf.own_layers().cpu().numpy().copy(), own=kf.own, maps=maps, poses=poses)
    ctx.close()


def _worker(rank, world, port, track, path, blank_at=None):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = {}
        run_loop(dist, track, out, blank_at)
        np.savez(path % rank, volume=out["volume"], own=np.int64(out["own"]), fused=np.int64(out["fused"]),
                 frame_counter=np.int64(out["frame_counter"]),
                 points=np.stack([p for p, _ in out["maps"]]), normals=np.stack([n for _, n in out["maps"]]),
                 pose_R=np.stack([r for r, _, _ in out["poses"]]), pose_t=np.stack([t for _, t, _ in out["poses"]]))
    finally:
        dist.destroy_process_group()


def spawn(world, track, tmp_path, blank_at=None):
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    path = str(tmp_path / ("w%d_t%d_r%%d.npz" % (world, int(track))))
    mp.spawn(_worker, args=(world, port, track, path, blank_at), nprocs=world, join=True)
    return [np.load(path % r) for r in range(world)]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_loop_with_supplied_poses_equals_the_single_gpu_loop(tmp_path, world):
    single = {}
    run_loop(None, False, single)
    ranks = spawn(world, False, tmp_path)
    assert [tuple(r["own"]) for r in ranks][0][0] == 0 and ranks[-1]["own"][1] == N
    whole = np.concatenate([r["volume"] for r in ranks])
    assert np.array_equal(whole, single["volume"])                              # slabs concatenate to the volume
    for r in ranks:                                                             # every rank holds the whole model maps
        for f in range(1, FRAMES):
            assert np.array_equal(r["points"][f].view(np.uint32), single["maps"][f][0].view(np.uint32)), (f,)
            assert np.array_equal(r["normals"][f].view(np.uint32), single["maps"][f][1].view(np.uint32)), (f,)
    assert (~np.isnan(single["maps"][-1][0][..., 0])).sum() > 50000


def test_sharded_loop_tracks_like_the_single_gpu_loop(tmp_path):
    single = {}
    run_loop(None, True, single)
    ranks = spawn(2, True, tmp_path)
    dims_vs = 3.0 / N
    for f in range(FRAMES):
        R1, t1, t_true = single["poses"][f]
        for r in ranks:
            assert np.array_equal(r["pose_R"][f], ranks[0]["pose_R"][f]) and np.array_equal(r["pose_t"][f], ranks[0]["pose_t"][f])
        assert np.abs(ranks[0]["pose_t"][f] - t1).max() < 2e-4                  # metres: rounding of the sums only
        assert np.abs(ranks[0]["pose_R"][f] - R1).max() < 2e-4
        assert np.linalg.norm(ranks[0]["pose_t"][f] - t_true) < 0.5 * dims_vs  # and both stay on the orbit


def test_lost_track_resets_every_rank_like_the_single_gpu_loop(tmp_path):
    """kinfu.cpp:247-252: when ICP fails the volume is cleared, the pose list reset and the frame counter set to 0,
    so the next frame seeds the model again.  A frame without returns in mid-sequence: the sharded loop and the
    one-rank loop report the same frames as fused, count frames alike, and end with the same volume."""
    blank = 3
    single = {}
    run_loop(None, True, single, blank_at=blank)
    assert single["fused"][blank] is False and single["frame_counter"][blank] == 0
    assert single["fused"][blank + 1] is False and single["frame_counter"][blank + 1] == 1   # the re-seeding frame
    assert single["fused"][blank + 2] is True
    ranks = spawn(2, True, tmp_path, blank_at=blank)
    for r in ranks:
        assert list(r["fused"]) == [int(v) for v in single["fused"]]
        assert list(r["frame_counter"]) == single["frame_counter"]
    whole = np.concatenate([r["volume"] for r in ranks])
    # after the reset both loops restart from the same first pose; tracking then differs by rounding only
    same = np.count_nonzero((whole >> 16) == (single["volume"] >> 16)) / whole.size
    assert same > 0.999, same
