"""The scenarios of the end-to-end loop tests (tests/test_oracle_loop.py on the CPU, tests/test_loop_vs_oracle_gpu.py on
the GPU): a 96^3 volume over the 3 m cube, frames of the S2 orbit at 640 x 480 with 1 mm noise.
  track   ten consecutive frames tracked from depth alone (identity hint, no gate)
  gate    hints = the true motions, tsdf_min_camera_movement between a one-frame and a two-frame step: only the frames
          that follow a skipped one are integrated (kinfu.cpp:279-290 looks at the hint); most do, so the model stays good
  reset   a frame without a single return in the middle: the ICP's determinant test fails, volume and pose are reset
          (kinfu.cpp:247-252), the next frame seeds the model again"""
import numpy as np

from yak_amd import synth

N = 96
SCENARIOS = {
    "track": dict(frames=list(range(10)), hints=False, min_move=0.0, blank=()),
    "gate": dict(frames=[0, 2, 4, 5, 7, 9, 11, 12, 14, 16], hints=True, min_move=0.06, blank=()),
    "reset": dict(frames=list(range(8)), hints=False, min_move=0.0, blank=(4,)),
}


def c2v(R_v2c, t_v2c):
    R = R_v2c.astype(np.float64).T
    return R, -R @ t_v2c.astype(np.float64)


def frames_of(name):
    sc = SCENARIOS[name]
    out = []
    for i, f in enumerate(sc["frames"]):
        depth, R, t = synth.scene_s2(f, noise_mm=1.0)
        if i in sc["blank"]:
            depth = np.zeros_like(depth)
        out.append((depth, c2v(R, t)))
    return out


def hint_of(prev_pose, cur_pose):
    """current -> previous camera: prev^-1 * cur"""
    Rp, tp = prev_pose
    Rc, tc = cur_pose
    return Rp.T @ Rc, Rp.T @ (tc - tp)


def run_oracle(name):
    from oracle.kinfu_loop import OracleKinFu

    sc = SCENARIOS[name]
    dims, vs, tr = synth.volume_geometry(N)
    fr = frames_of(name)
    kf = OracleKinFu(dims, vs, tr, synth.MAX_WEIGHT, synth.INTR, synth.ROWS, synth.COLS, volume_pose=fr[0][1],
                     tsdf_min_camera_movement=sc["min_move"], use_pose_hints=sc["hints"])
    poses, flags = [], []
    for i, (depth, true_pose) in enumerate(fr):
        hint = hint_of(fr[i - 1][1], true_pose) if sc["hints"] and i > 0 else None
        kf(depth, motion_hint=hint)
        poses.append(kf.pose)
        flags.append((kf.fused, kf.integrated, kf.reset))
    return {"poses": poses, "flags": flags, "weights_nonzero": int(np.count_nonzero(kf.volume >> 16)), "volume": kf.volume,
            "true": [p for _, p in fr]}
