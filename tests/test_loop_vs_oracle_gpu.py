"""The frame loop on the GPU against the reference's loop restated on the CPU (oracle/kinfu_loop.py; kinfu.cpp:162-309,
projective_icp.cpp:175-232), frame by frame on the scenarios of tests/loop_scenario.py: tracking from depth alone, the
movement gate on the hint, a reset in the middle.  GPU side: yak_amd/kinfu_sharded.py with one rank -- the product's
host mirror of KinFu::operator() over the C ABI (device-resident ICP schedule, integrate, ray cast).

Tolerances: poses within 1e-4 m and 1e-4 rad per frame -- the ICP sums are float32 in another order on the
GPU and its 6 x 6 solve is a double Cholesky, the CPU loop's a float64 SVD (both stand-ins for OpenCV's: parity unpinned) --
the gate's and the reset's decisions identical, voxel weights differing on less than 0.1 % of the voxels (a pose that
differs in the seventh digit moves a voxel across the band's edge now and then)."""
import numpy as np
import pytest
import torch

from loop_scenario import N, SCENARIOS, frames_of, hint_of, run_oracle
from yak_amd import synth
from yak_amd.kinfu_sharded import ShardedKinFu

pytestmark = pytest.mark.gpu


def _angle(Ra, Rb):
    # (|Ra^T Rb - I|_F / sqrt 2: the angle for small ones; arccos of the trace loses half the digits next to 1)
    return float(np.linalg.norm(Ra.T @ Rb - np.eye(3)) / np.sqrt(2.0))


@pytest.mark.parametrize("name", sorted(SCENARIOS))
def test_gpu_loop_follows_the_cpu_loop(gpu_ctx, name):
    sc = SCENARIOS[name]
    want = run_oracle(name)
    dims, vs, tr = synth.volume_geometry(N)
    fr = frames_of(name)
    kf = ShardedKinFu(gpu_ctx, None, dims, vs, tr, synth.MAX_WEIGHT, synth.INTR, synth.ROWS, synth.COLS,
                      tsdf_min_camera_movement=sc["min_move"])
    for i, (depth, true_pose) in enumerate(fr):
        d = torch.from_numpy(depth.view(np.int16)).cuda()
        hint = hint_of(fr[i - 1][1], true_pose) if sc["hints"] and i > 0 else None
        fused = kf(d, first_pose=fr[0][1] if kf.frame == 0 else None, motion_hint=hint)
        w_fused, w_integrated, w_reset = want["flags"][i]
        assert bool(fused) == w_fused and kf.integrated == w_integrated, (name, i)
        assert (kf.frame == 0) == w_reset, (name, i)                       # a reset leaves the frame counter at 0
        R, t = want["poses"][i]
        assert np.linalg.norm(kf.pose_t - t) < 1e-4 and _angle(kf.pose_R, R) < 1e-4, (
            name, i, np.linalg.norm(kf.pose_t - t), _angle(kf.pose_R, R))
    got = kf.own_layers().cpu().numpy().view(np.uint32)
    differ = np.count_nonzero((got >> 16) != (want["volume"] >> 16))
    assert differ < 1e-3 * got.size, "%s: %d of %d voxel weights differ" % (name, differ, got.size)
    assert np.count_nonzero(got >> 16) > 0
