"""The reference's frame loop restated end to end on the CPU (oracle/kinfu_loop.py; kinfu.cpp:162-309,
projective_icp.cpp:175-232): no vector in the reference pins it (PARITY UNPINNED), so it is pinned by what it must do --
recover the synthetic orbit's true poses, integrate only when the hint says the camera moved, reset on a frame without
returns and start again -- and guarded against drift by tests/golden/kinfu_loop_96.npz (tests/golden/gen_kinfu_loop.py)."""
import os

import numpy as np
import pytest

from loop_scenario import SCENARIOS, run_oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kinfu_loop_96.npz")


@pytest.fixture(scope="module")
def runs():
    return {name: run_oracle(name) for name in SCENARIOS}


def _angle(Ra, Rb):
    # (|Ra^T Rb - I|_F / sqrt 2: the angle for small ones; arccos of the trace loses half the digits next to 1)
    return float(np.linalg.norm(Ra.T @ Rb - np.eye(3)) / np.sqrt(2.0))


def test_tracking_recovers_the_true_orbit(runs):
    tr = runs["track"]
    assert [f[0] for f in tr["flags"]] == [False] + [True] * 9          # frame 0 reports false (kinfu.cpp:191-201)
    assert all(f[1] for f in tr["flags"]) and not any(f[2] for f in tr["flags"])
    for (R, t), (Rt, tt) in zip(tr["poses"], tr["true"]):
        assert np.linalg.norm(t - tt) < 0.02 and _angle(R, Rt) < 0.01   # 31 mm voxels, 1 mm noise: centimetres, not drift
    assert tr["weights_nonzero"] > 50000


def test_the_gate_looks_at_the_hint(runs):
    tr = runs["gate"]
    steps = np.diff(SCENARIOS["gate"]["frames"])
    want = [True] + [bool(s >= 2) for s in steps]                        # one orbit step: (0.021 + 0.067) / 2 = 0.044 < 0.06
    assert [f[1] for f in tr["flags"]] == want
    assert [f[0] for f in tr["flags"]] == [False] + [True] * 9           # fused (tracked, ray cast) whether integrated or not
    for (R, t), (Rt, tt) in zip(tr["poses"], tr["true"]):
        assert np.linalg.norm(t - tt) < 0.02 and _angle(R, Rt) < 0.01


def test_a_frame_without_returns_resets_volume_and_pose(runs):
    tr = runs["reset"]
    assert [f[2] for f in tr["flags"]] == [False, False, False, False, True, False, False, False]
    assert [f[0] for f in tr["flags"]] == [False, True, True, True, False, False, True, True]   # frame 5 seeds the model again
    first = tr["poses"][0]
    assert np.array_equal(tr["poses"][4][0], first[0]) and np.array_equal(tr["poses"][4][1], first[1])  # resetPose: volume_pose


def test_against_the_committed_trajectory(runs):
    g = np.load(GOLDEN)
    for name, tr in runs.items():
        assert np.array_equal(g[name + "_flags"], np.array(tr["flags"], dtype=np.uint8)), name
        assert np.abs(g[name + "_R"] - np.stack([r for r, _ in tr["poses"]])).max() < 1e-6, name
        assert np.abs(g[name + "_t"] - np.stack([t for _, t in tr["poses"]])).max() < 1e-6, name
        assert int(g[name + "_weights_nonzero"][0]) == tr["weights_nonzero"], name
