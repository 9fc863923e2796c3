"""The KinFu frame loop over a z-slab sharded volume, one process per GPU (BASELINE config 4).

Mirrors kfusion::KinFu::operator() (reference yak/src/kfusion/kinfu.cpp:162-309) call for call, as
yak_amd/host/kinfu.cpp does for one GPU; what changes with more than one rank:

  * every rank holds its own layers plus the ray-cast halo and integrates all of them (the halo
    redundantly: integrate is a per-voxel function of the frame, so nothing is exchanged);
  * pre-processing of the depth frame (bilateral, pyramid, point / normal maps) is replicated -- it is
    image-space and costs 0.08 ms;
  * ICP: every rank accumulates the sums of its band of rows, the 33 values are all-reduced, every rank
    solves the same 6 x 6 system (projective_icp.cpp:206-229) and so holds the same pose;
  * raycast: yakhip_raycast_slab -> composite (two all-reduces) -> yakhip_raycast_resolve -> the few
    deferred pixels once more; every rank ends up with the whole model maps.

This is host-side plumbing over the C ABI (yak_amd/capi.py) and torch.distributed; all arithmetic on
images and voxels happens in libyakhip.  The reference has no multi-GPU mode: there is nothing to cite
for the collectives.
"""
import math

import numpy as np

from . import capi
from .shard import composite_raycast, held_range, row_range, slab_range

LEVELS = 3                     # ProjectiveICP::MAX_PYRAMID_LEVELS - 1 used levels, kinfu.cpp:10-48 (iters 10, 5, 4, 0)
ITERATIONS = (10, 5, 4)        # on levels 0, 1, 2 as ProjectiveICP::setIterationsNum stores them (projective_icp.cpp:87-96)


def rodrigues(r):
    """cv::Affine3f(rvec, tvec)'s rotation, projective_icp.cpp:225."""
    th = float(np.linalg.norm(r))
    if th < 1e-12:
        return np.eye(3)
    k = np.asarray(r, dtype=np.float64) / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * (K @ K)


def normal_equations(sums):
    """StreamHelper::get, projective_icp.cpp:42-61: 27 sums -> symmetric A (6 x 6), b (6)."""
    A, b, k = np.zeros((6, 6)), np.zeros(6), 0
    for i in range(6):
        for j in range(i, 7):
            if j == 6:
                b[i] = sums[k]
            else:
                A[i, j] = A[j, i] = sums[k]
            k += 1
    return A, b


class ShardedKinFu:
    def __init__(self, ctx, dist, dims, voxel_size, trunc_dist, max_weight, intr, rows, cols, device="cuda",
                 bilateral=(7, 4.5, 0.04), icp_dist_thres=0.1, icp_angle_thres=math.radians(30.0), step_factor=0.75,
                 gradient_delta_factor=0.5):
        import torch

        self.torch, self.ctx, self.dist = torch, ctx, dist
        self.world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        self.dims, self.vs, self.trunc, self.max_weight = tuple(int(d) for d in dims), tuple(voxel_size), trunc_dist, max_weight
        self.intr, self.rows, self.cols, self.device = tuple(intr), rows, cols, device
        self.bilateral, self.dist_thres, self.angle_thres = bilateral, icp_dist_thres, icp_angle_thres
        self.step_factor, self.gdf = step_factor, gradient_delta_factor

        self.own = slab_range(self.dims[2], self.world, self.rank)
        probe = capi.volume_desc(256, self.dims, self.vs, trunc_dist, max_weight)
        self.halo = capi.raycast_halo(probe, step_factor, gradient_delta_factor) if self.world > 1 else 0
        self.held = held_range(self.dims[2], self.own[0], self.own[1], self.halo)
        plane = self.dims[0] * self.dims[1]
        self.volume = torch.zeros(plane * (self.held[1] - self.held[0]), dtype=torch.int32, device=device)
        self.desc = capi.volume_desc(self.volume.data_ptr(), self.dims, self.vs, trunc_dist, max_weight, *self.held)
        ctx.clear_volume(self.desc)

        i16 = lambda r, c: torch.zeros((r, c), dtype=torch.int16, device=device)        # noqa: E731
        f4 = lambda r, c: torch.zeros((r, c, 4), dtype=torch.float32, device=device)    # noqa: E731
        lv = [(rows >> l, cols >> l) for l in range(LEVELS)]
        self.dists = i16(rows, cols)
        self.cur_d = [i16(r, c) for r, c in lv]
        self.cur_p, self.cur_n = [f4(r, c) for r, c in lv], [f4(r, c) for r, c in lv]
        self.prv_p, self.prv_n = [f4(r, c) for r, c in lv], [f4(r, c) for r, c in lv]
        self.keys = torch.zeros((rows, cols), dtype=torch.int32, device=device)
        self.sums = torch.zeros(33, dtype=torch.float32, device=device)
        self.pose_R, self.pose_t = np.eye(3), np.zeros(3)          # camera -> volume
        self.frame = 0

    # ---- per-frame stages -------------------------------------------------------------------------
    def _preprocess(self, depth):
        ctx, rows, cols = self.ctx, self.rows, self.cols
        k, ss, sd = self.bilateral
        ctx.compute_dists(depth.data_ptr(), cols * 2, self.dists.data_ptr(), cols * 2, rows, cols, self.intr)   # kinfu.cpp:171
        ctx.bilateral_filter(depth.data_ptr(), cols * 2, self.cur_d[0].data_ptr(), cols * 2, rows, cols, k, ss, sd)
        for l in range(1, LEVELS):
            ctx.depth_pyramid(self.cur_d[l - 1].data_ptr(), (cols >> (l - 1)) * 2, rows >> (l - 1), cols >> (l - 1),
                              self.cur_d[l].data_ptr(), (cols >> l) * 2, sd)
        for l in range(LEVELS):
            li = tuple(np.float32(v) / np.float32(1 << l) for v in self.intr)
            ctx.compute_points_normals(li, self.cur_d[l].data_ptr(), (cols >> l) * 2, rows >> l, cols >> l,
                                       self.cur_p[l].data_ptr(), (cols >> l) * 16, self.cur_n[l].data_ptr(), (cols >> l) * 16)

    def _estimate_motion(self):
        """ProjectiveICP::estimateTransform, projective_icp.cpp:175-232, with the sums of each linearisation
        gathered over the ranks.  Returns (ok, R, t): current -> previous camera."""
        R, t = np.eye(3), np.zeros(3)
        for l in reversed(range(LEVELS)):
            r_, c_ = self.rows >> l, self.cols >> l
            r0, r1 = row_range(r_, self.world, self.rank)
            for _ in range(ITERATIONS[l]):
                self.ctx.icp_accumulate_rows(np.float32(R).reshape(-1), np.float32(t), self.intr, l, self.dist_thres,
                                             self.angle_thres, self.cur_p[l].data_ptr(), c_ * 16, self.cur_n[l].data_ptr(),
                                             c_ * 16, self.prv_p[l].data_ptr(), c_ * 16, self.prv_n[l].data_ptr(), c_ * 16,
                                             r_, c_, r0, r1, self.sums.data_ptr())
                self.ctx.synchronize()
                total = _all_reduce(self.dist, self.sums, "sum").cpu().numpy().astype(np.float64)
                A, b = normal_equations(total[:27])
                det = np.linalg.det(A)
                if not (abs(det) >= 1e-15):                         # projective_icp.cpp:213 (NaN fails too)
                    return False, R, t
                x = np.linalg.solve(A, b)
                Ri = rodrigues(x[:3])
                R, t = Ri @ R, Ri @ t + x[3:]                       # :225-227  Tinc * (R, t)
        return True, R, t

    def _integrate(self):
        Rv = self.pose_R.T
        tv = -Rv @ self.pose_t
        self.ctx.integrate(self.desc, self.dists.data_ptr(), self.cols * 2, self.rows, self.cols, np.float32(Rv).reshape(-1),
                           np.float32(tv), self.intr)

    def _raycast_model(self):
        """TsdfVolume::raycast at the current pose into the level-0 model maps of every rank, then the pyramid."""
        ctx, rows, cols = self.ctx, self.rows, self.cols
        R, t = np.float32(self.pose_R).reshape(-1), np.float32(self.pose_t)
        Rinv = np.float32(self.pose_R.T).reshape(-1)
        if self.world == 1:
            ctx.raycast_points(self.desc, R, t, Rinv, self.intr, self.prv_p[0].data_ptr(), cols * 16, self.prv_n[0].data_ptr(),
                               cols * 16, rows, cols, self.step_factor, self.gdf)
        else:
            ctx.raycast_slab(self.desc, self.own[0], self.own[1], R, t, Rinv, self.intr, self.keys.data_ptr(), cols * 4,
                             self.prv_n[0].data_ptr(), cols * 16, rows, cols, points_ptr=self.prv_p[0].data_ptr(),
                             points_pitch=cols * 16, step_factor=self.step_factor, gradient_delta_factor=self.gdf)
            ctx.synchronize()

            def resolve(keys, images):
                ctx.raycast_resolve(self.desc, self.own[0], self.own[1], Rinv, keys.data_ptr(), cols * 4, images[1].data_ptr(),
                                    cols * 16, rows, cols, points_ptr=images[0].data_ptr(), points_pitch=cols * 16,
                                    gradient_delta_factor=self.gdf)
                ctx.synchronize()

            (p0, n0), best, left = composite_raycast(self.dist, self.keys, [self.prv_p[0], self.prv_n[0]],
                                                     [float("nan"), float("nan")], resolve=resolve,
                                                     all_reduce=_all_reduce)
            assert left == 0
            self.prv_p[0].copy_(p0)
            self.prv_n[0].copy_(n0)
        for l in range(1, LEVELS):
            ctx.resize_points_normals(self.prv_p[l - 1].data_ptr(), (cols >> (l - 1)) * 16, self.prv_n[l - 1].data_ptr(),
                                      (cols >> (l - 1)) * 16, rows >> (l - 1), cols >> (l - 1), self.prv_p[l].data_ptr(),
                                      (cols >> l) * 16, self.prv_n[l].data_ptr(), (cols >> l) * 16)

    # ---- the loop body ----------------------------------------------------------------------------
    def __call__(self, depth, pose=None, first_pose=None):
        """depth: int16 (rows, cols) device tensor of millimetres.  pose = (R, t) camera -> volume to use
        instead of tracking (kinfu.cpp:228-253 with use_icp off); first_pose: where frame 0 is integrated.
        Returns True when the frame was fused (the reference reports false for frame 0, :191-201)."""
        self._preprocess(depth)
        if self.frame == 0:
            if first_pose is not None:
                self.pose_R, self.pose_t = np.asarray(first_pose[0], dtype=np.float64), np.asarray(first_pose[1], dtype=np.float64)
            self._integrate()
            # kinfu.cpp:196-198 swaps the frame's own maps in as the first model maps
            self.cur_p, self.prv_p = self.prv_p, self.cur_p
            self.cur_n, self.prv_n = self.prv_n, self.cur_n
            self.ctx.synchronize()
            self.frame += 1
            return False
        if pose is None:
            ok, R, t = self._estimate_motion()
            if not ok:
                return False
            self.pose_R, self.pose_t = self.pose_R @ R, self.pose_R @ t + self.pose_t     # kinfu.cpp:256-268
        else:
            self.pose_R, self.pose_t = np.asarray(pose[0], dtype=np.float64), np.asarray(pose[1], dtype=np.float64)
        self._integrate()
        self._raycast_model()
        self.ctx.synchronize()
        self.frame += 1
        return True

    def own_layers(self):
        """This rank's own layers of the volume (a view into the held tensor)."""
        plane = self.dims[0] * self.dims[1]
        a = (self.own[0] - self.held[0]) * plane
        return self.volume[a:a + (self.own[1] - self.own[0]) * plane]


def _all_reduce(dist, tensor, op):
    """all_reduce that also works for device tensors over a CPU-only backend (gloo in the tests)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return tensor
    red = {"sum": dist.ReduceOp.SUM, "min": dist.ReduceOp.MIN}[op]
    if tensor.is_cuda and dist.get_backend() == "gloo":
        host = tensor.cpu()
        dist.all_reduce(host, op=red)
        tensor.copy_(host)
    else:
        dist.all_reduce(tensor, op=red)
    return tensor
