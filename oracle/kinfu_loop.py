"""TEST INFRASTRUCTURE (oracle): the reference's frame loop restated on the CPU, end to end.

kfusion::KinFu::operator() (reference yak/src/kfusion/kinfu.cpp:162-309) over the scalar restatements of
oracle/yak_oracle.c -- compute_dists, bilateral filter, depth pyramid, point / normal maps, projective ICP, integrate,
ray cast, 2 x 2 resize -- with ProjectiveICP::estimateTransform (yak/src/kfusion/projective_icp.cpp:175-232) as the
host loop it is there: one linearisation at a time, determinant test, 6 x 6 solve, Rodrigues increment composed from
the left.  Kept: the frame-0 rule (:191-201: integrate at the pose on top of the list, swap the pyramids, report false),
the movement gate on the HINT motion, not the corrected one (:279-285), reset of volume and pose on a failed ICP
(:247-252), poses_.back() * cameraMotionCorrected with update_via_sensor_motion (:256-268).

PARITY UNPINNED: the reference holds no trajectory, and its 6 x 6 solve is OpenCV's (cv::determinant, cv::solve
DECOMP_SVD, cv::Rodrigues: un-vendored); numpy's float64 SVD solve stands in.  What pins this loop is
physics: it recovers the synthetic orbit's true poses (tests/test_oracle_loop.py), and tests/golden/kinfu_loop_96.npz
guards it against drift.  Only tests import this file.
"""
import math

import numpy as np

from . import (bilateral_filter, compute_dists, compute_points_normals, depth_pyramid, icp_accumulate, integrate, raycast,
               resize_points_normals, truncate_depth)


def rodrigues(r):
    """cv::Affine3f(rvec, t)'s rotation (cv::Rodrigues), in float64, handed on as float32"""
    r = np.asarray(r, dtype=np.float64)
    th = float(np.linalg.norm(r))
    if th < 1e-12:
        return np.eye(3)
    k = r / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * K @ K


def rvec_norm(R):
    """cv::norm(Affine3f::rvec()): the rotation angle"""
    c = (np.trace(np.asarray(R, dtype=np.float64)) - 1.0) / 2.0
    return float(math.acos(min(1.0, max(-1.0, c))))


class OracleKinFu:
    """Poses are camera -> volume as (R 3x3, t 3) float64 pairs rounded through float32 where the reference holds
    Affine3f.  After every call: .fused (the return value), .integrated (the gate), .reset (a failed ICP)."""

    def __init__(self, dims, voxel_size, trunc_dist, max_weight, intr, rows, cols, volume_pose=None, iters=(10, 5, 4),
                 bilateral=(7, 4.5, 0.04), icp_truncate_depth_dist=0.0, icp_dist_thres=0.1, icp_angle_thres=math.radians(30.0),
                 step_factor=0.75, gradient_delta_factor=0.5, tsdf_min_camera_movement=0.0, use_pose_hints=False, use_icp=True,
                 update_via_sensor_motion=True):
        self.dims, self.vs, self.trunc, self.max_weight = tuple(int(d) for d in dims), tuple(voxel_size), trunc_dist, max_weight
        self.intr, self.rows, self.cols = tuple(intr), rows, cols
        self.iters, self.bilateral, self.trunc_depth = tuple(iters), bilateral, icp_truncate_depth_dist
        self.dist_thres, self.angle_thres = icp_dist_thres, icp_angle_thres
        self.step_factor, self.gdf, self.min_move = step_factor, gradient_delta_factor, tsdf_min_camera_movement
        self.use_pose_hints, self.use_icp, self.via_motion = use_pose_hints, use_icp, update_via_sensor_motion
        self.volume_pose = (np.eye(3), np.zeros(3)) if volume_pose is None else (np.array(volume_pose[0], float), np.array(volume_pose[1], float))
        self.volume = np.zeros(self.dims[0] * self.dims[1] * self.dims[2], dtype=np.uint32)
        self.poses = [self.volume_pose]          # kinfu.cpp:50-80: poses_ starts with params.volume_pose
        self.frame = 0
        self.prev_p = self.prev_n = None
        self.fused = self.integrated = self.reset = False

    # ---- pieces -----------------------------------------------------------------------------------
    def _level_intr(self, l):
        # Intr::operator()(level), precomp.cpp:10-14: every entry divided by 2^level in float
        return tuple(float(np.float32(v) / np.float32(1 << l)) for v in self.intr)

    def _integrate(self, dists, pose):
        R_c2v, t_c2v = pose
        Rv = np.float32(R_c2v.T)                           # tsdf_volume.cpp:97: vol2cam = camera_pose.inv() * pose_ (identity)
        tv = np.float32(-(R_c2v.T @ t_c2v))
        integrate(self.volume, self.dims, self.vs, self.trunc, self.max_weight, dists, Rv, tv, self.intr)

    def estimate_transform(self, R0, t0, cur_p, cur_n):
        """projective_icp.cpp:175-232.  Returns (ok, R, t): current -> previous camera."""
        R, t = np.float32(R0).astype(np.float64), np.float32(t0).astype(np.float64)
        for level in range(len(self.iters) - 1, -1, -1):
            for _ in range(self.iters[level]):
                sums, _ = icp_accumulate(R, t, self.intr, level, self.dist_thres, self.angle_thres, cur_p[level], cur_n[level],
                                         self.prev_p[level], self.prev_n[level])
                s = np.float32(sums).astype(np.float64)    # the reference's buffer is float (proj_icp.cu:366-387)
                A, b, k = np.zeros((6, 6)), np.zeros(6), 0
                for i in range(6):                         # StreamHelper::get, projective_icp.cpp:50-59
                    for j in range(i, 7):
                        if j == 6:
                            b[i] = s[k]
                        else:
                            A[i, j] = A[j, i] = s[k]
                        k += 1
                det = np.linalg.det(A)
                if abs(det) < 1e-15 or np.isnan(det):      # :215-222
                    return False, R, t
                r = np.float32(np.linalg.lstsq(A, b, rcond=None)[0]).astype(np.float64)   # cv::solve(DECOMP_SVD), :225
                Rinc = np.float32(rodrigues(r[:3])).astype(np.float64)
                R = np.float32(Rinc @ R).astype(np.float64)                               # Tinc * correctedAffine, :228
                t = np.float32(Rinc @ t + r[3:]).astype(np.float64)
        return True, R, t

    # ---- kinfu.cpp:162-309 ------------------------------------------------------------------------
    def __call__(self, depth, motion_hint=None, current_pose=None, previous_pose=None):
        """depth: uint16 (rows, cols) millimetres.  motion_hint (R, t): inputCameraMotion, current -> previous camera;
        current_pose / previous_pose: the poses the caller believes in (used as the reference uses them)."""
        self.fused = self.integrated = self.reset = False
        dists = compute_dists(depth, self.intr)                                           # :171
        k, ss, sd = self.bilateral
        cur_d = [bilateral_filter(depth, k, ss, sd)]                                      # :172-173
        if self.trunc_depth > 0:
            cur_d[0] = truncate_depth(cur_d[0], self.trunc_depth)                         # :175-176
        for _ in range(1, len(self.iters)):
            cur_d.append(depth_pyramid(cur_d[-1], sd))                                    # :178-179
        cur_p, cur_n = [], []
        for l, d in enumerate(cur_d):
            p, n = compute_points_normals(d, self._level_intr(l))                         # :185
            cur_p.append(p)
            cur_n.append(n)
        if self.frame == 0:                                                               # :191-201
            self._integrate(dists, self.poses[-1])
            self.integrated = True
            self.prev_p, self.prev_n = cur_p, cur_n
            self.frame += 1
            return False
        ident = (np.eye(3), np.zeros(3))
        motion = motion_hint if (self.use_pose_hints and motion_hint is not None) else ident           # :213-219
        if self.use_icp:
            ok, Rm, tm = self.estimate_transform(motion[0], motion[1], cur_p, cur_n)                  # :229-238
            prev = previous_pose if previous_pose is not None else self.poses[-1]
            pose_corr = (prev[0] @ Rm, prev[0] @ tm + prev[1])
        else:
            ok, (Rm, tm) = True, motion
            pose_corr = current_pose if current_pose is not None else self.poses[-1]
        if not ok:                                                                        # :247-252
            self.volume[:] = 0
            self.frame = 0
            self.poses = [self.volume_pose]
            self.reset = True
            return False
        if self.via_motion:                                                               # :256-268
            b = self.poses[-1]
            self.poses.append((np.float32(b[0] @ Rm).astype(np.float64), np.float32(b[0] @ tm + b[1]).astype(np.float64)))
        else:
            self.poses.append(pose_corr)
        rnorm, tnorm = np.float32(rvec_norm(motion[0])), np.float32(np.linalg.norm(motion[1]))       # :279-280, the HINT
        if (rnorm + tnorm) / 2 >= self.min_move:                                          # :285-290
            self._integrate(dists, self.poses[-1])
            self.integrated = True
        R, t = self.poses[-1]                                                             # :301-303
        p0, n0, _, _ = raycast(self.volume, self.dims, self.vs, self.trunc, np.float32(R), np.float32(t), np.float32(R.T), self.intr,
                         self.rows, self.cols, self.step_factor, self.gdf)
        self.prev_p, self.prev_n = [p0], [n0]
        for _ in range(1, len(self.iters)):
            p, n = resize_points_normals(self.prev_p[-1], self.prev_n[-1])
            self.prev_p.append(p)
            self.prev_n.append(n)
        self.frame += 1
        self.fused = True
        return True

    @property
    def pose(self):
        return self.poses[-1]
