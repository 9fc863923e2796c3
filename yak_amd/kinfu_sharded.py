"""The KinFu frame loop over a z-slab sharded volume, one process per GPU (BASELINE config 4).

Mirrors kfusion::KinFu::operator() (reference yak/src/kfusion/kinfu.cpp:162-309) call for call, as
yak_amd/host/kinfu.cpp does for one GPU; what changes with more than one rank:

  * every rank holds its own layers plus the ray-cast halo and integrates all of them (the halo
    redundantly: integrate is a per-voxel function of the frame, so nothing is exchanged);
  * pre-processing of the depth frame (bilateral, pyramid, point / normal maps) is replicated -- it is
    image-space and costs 0.08 ms;
  * ICP: yakhip_icp_estimate_sharded -- every rank accumulates the sums of its band of rows, the 33 values
    of each linearisation are all-reduced on the stream, every rank solves the same 6 x 6 system on the device
    (projective_icp.cpp:206-229) and so holds the same pose; the host waits once per frame;
  * raycast: yakhip_raycast_sharded -- slab march, composite (two all-reduces), resolve, the few deferred
    pixels once more (a third); every rank ends up with the whole model maps;
  * a lost track resets volume, pose and frame counter on every rank alike (kinfu.cpp:247-252).

This is host-side plumbing over the C ABI (yak_amd/capi.py); all arithmetic on images and voxels AND the
collectives of the frame loop happen in libyakhip (RCCL when torch.distributed runs over nccl; over gloo the
library calls back into make_comm's host-staged all-reduce).  The C++ counterpart is kfusion::ShardedKinFu
(yak_amd/host/kinfu_sharded.cpp).  The reference has no multi-GPU mode: there is nothing to cite
for the collectives.
"""
import math

import numpy as np

from . import capi
from .shard import held_range, slab_range

LEVELS = 3                     # ProjectiveICP::MAX_PYRAMID_LEVELS - 1 used levels, kinfu.cpp:10-48 (iters 10, 5, 4, 0)
ITERATIONS = (10, 5, 4)        # on levels 0, 1, 2 as ProjectiveICP::setIterationsNum stores them (projective_icp.cpp:87-96)


class ShardedKinFu:
    def __init__(self, ctx, dist, dims, voxel_size, trunc_dist, max_weight, intr, rows, cols, device="cuda",
                 bilateral=(7, 4.5, 0.04), icp_dist_thres=0.1, icp_angle_thres=math.radians(30.0), step_factor=0.75,
                 gradient_delta_factor=0.5, comm=None, tsdf_min_camera_movement=0.0):
        import torch

        self.torch, self.ctx, self.dist = torch, ctx, dist
        self.world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        self.dims, self.vs, self.trunc, self.max_weight = tuple(int(d) for d in dims), tuple(voxel_size), trunc_dist, max_weight
        self.intr, self.rows, self.cols, self.device = tuple(intr), rows, cols, device
        self.bilateral, self.dist_thres, self.angle_thres = bilateral, icp_dist_thres, icp_angle_thres
        self.step_factor, self.gdf = step_factor, gradient_delta_factor
        self.min_move = tsdf_min_camera_movement   # KinFuParams::tsdf_min_camera_movement (0: every frame is integrated)
        self.integrated = False                    # whether the last call integrated its frame (the movement gate)

        self.own = slab_range(self.dims[2], self.world, self.rank)
        probe = capi.volume_desc(256, self.dims, self.vs, trunc_dist, max_weight)
        self.halo = capi.raycast_halo(probe, step_factor, gradient_delta_factor) if self.world > 1 else 0
        self.held = held_range(self.dims[2], self.own[0], self.own[1], self.halo)
        plane = self.dims[0] * self.dims[1]
        self.volume = torch.zeros(plane * (self.held[1] - self.held[0]), dtype=torch.int32, device=device)
        self.desc = capi.volume_desc(self.volume.data_ptr(), self.dims, self.vs, trunc_dist, max_weight, *self.held)
        # the surface map of the held layers: kept by integrate, consulted by the slab march (include/yakhip.h)
        self.surface_map = torch.zeros(capi.surface_map_bytes(self.desc) // 4, dtype=torch.int32, device=device)
        self.desc = capi.volume_desc(self.volume.data_ptr(), self.dims, self.vs, trunc_dist, max_weight, *self.held,
                                     surface_map=self.surface_map.data_ptr())
        ctx.clear_volume(self.desc)

        i16 = lambda r, c: torch.zeros((r, c), dtype=torch.int16, device=device)        # noqa: E731
        f4 = lambda r, c: torch.zeros((r, c, 4), dtype=torch.float32, device=device)    # noqa: E731
        lv = [(rows >> l, cols >> l) for l in range(LEVELS)]
        self.dists = i16(rows, cols)
        self.cur_d = [i16(r, c) for r, c in lv]
        self.cur_p, self.cur_n = [f4(r, c) for r, c in lv], [f4(r, c) for r, c in lv]
        self.prv_p, self.prv_n = [f4(r, c) for r, c in lv], [f4(r, c) for r, c in lv]
        self.keys = torch.zeros((rows, cols), dtype=torch.int32, device=device)
        self.comm = comm if comm is not None else make_comm(ctx, dist)
        self.first_R, self.first_t = np.eye(3), np.zeros(3)
        self.pose_R, self.pose_t = np.eye(3), np.zeros(3)          # camera -> volume
        self.frame = 0
        self.stage_events = None  # measurement aid: a list to collect (stage, start event, end event) per call (bench.py)

    def _stage(self, name, fn):
        """Run one stage of the frame; with stage_events set, between two events on torch's current stream (the
        context launches there)."""
        if self.stage_events is None:
            return fn()
        e0, e1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        self.stage_events.append((name, e0, e1))
        return r

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

    def _estimate_motion(self, hint=None):
        """ProjectiveICP::estimateTransform, projective_icp.cpp:175-232, over the ranks: yakhip_icp_estimate_sharded
        runs the whole coarse-to-fine schedule on the device -- every rank its band of rows, the 33 sums of each
        linearisation all-reduced on the stream -- and the host waits once.  hint (R, t): the motion the estimate
        starts from (use_pose_hints, kinfu.cpp:213-219; identity otherwise).  Returns (ok, R, t): current ->
        previous camera, the same bits on every rank."""
        levels = [(self.cur_p[l].data_ptr(), self.cur_n[l].data_ptr(), self.prv_p[l].data_ptr(), self.prv_n[l].data_ptr(),
                   (self.cols >> l) * 16, self.rows >> l, self.cols >> l, ITERATIONS[l]) for l in range(LEVELS)]
        R0, t0 = (np.eye(3), np.zeros(3)) if hint is None else hint
        R, t, ok = self.ctx.icp_estimate_sharded(self.comm, np.float32(R0).reshape(-1), np.float32(t0), self.intr, self.dist_thres,
                                                 self.angle_thres, levels)
        return ok, np.float64(R).reshape(3, 3), np.float64(t)

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
            # slab march, composite (MIN over the keys, SUM over the winners' payload), resolve, composite of the
            # settled pixels: one call, enqueued on the stream, nothing waited for (include/yakhip.h)
            ctx.raycast_sharded(self.comm, self.desc, self.own[0], self.own[1], R, t, Rinv, self.intr, self.keys.data_ptr(),
                                cols * 4, self.prv_n[0].data_ptr(), cols * 16, rows, cols, points_ptr=self.prv_p[0].data_ptr(),
                                points_pitch=cols * 16, step_factor=self.step_factor, gradient_delta_factor=self.gdf)
        for l in range(1, LEVELS):
            ctx.resize_points_normals(self.prv_p[l - 1].data_ptr(), (cols >> (l - 1)) * 16, self.prv_n[l - 1].data_ptr(),
                                      (cols >> (l - 1)) * 16, rows >> (l - 1), cols >> (l - 1), self.prv_p[l].data_ptr(),
                                      (cols >> l) * 16, self.prv_n[l].data_ptr(), (cols >> l) * 16)

    # ---- the loop body ----------------------------------------------------------------------------
    def __call__(self, depth, pose=None, first_pose=None, motion_hint=None):
        """depth: int16 (rows, cols) device tensor of millimetres.  pose = (R, t) camera -> volume to use
        instead of tracking (kinfu.cpp:228-253 with use_icp off); first_pose: where frame 0 is integrated;
        motion_hint (R, t): inputCameraMotion with use_pose_hints (current -> previous camera) -- where the ICP
        starts, and what the movement gate looks at (kinfu.cpp:279-285: the hint, not the corrected motion).
        Returns True when the frame was fused (the reference reports false for frame 0, :191-201)."""
        self._stage("preprocess", lambda: self._preprocess(depth))
        self.integrated = False
        if self.frame == 0:
            if first_pose is not None:
                self.pose_R, self.pose_t = np.asarray(first_pose[0], dtype=np.float64), np.asarray(first_pose[1], dtype=np.float64)
                self.first_R, self.first_t = self.pose_R.copy(), self.pose_t.copy()
            self._integrate()
            self.integrated = True
            # kinfu.cpp:196-198 swaps the frame's own maps in as the first model maps
            self.cur_p, self.prv_p = self.prv_p, self.cur_p
            self.cur_n, self.prv_n = self.prv_n, self.cur_n
            self.ctx.synchronize()
            self.frame += 1
            return False
        if pose is None:
            ok, R, t = self._stage("icp", lambda: self._estimate_motion(motion_hint))
            if not ok:
                # kinfu.cpp:247-252: resetVolume() + resetPose() -- the volume is cleared, the frame counter goes
                # to 0 and the next frame seeds the model again (every rank takes this branch alike)
                self.ctx.clear_volume(self.desc)
                self.pose_R, self.pose_t = self.first_R.copy(), self.first_t.copy()
                self.ctx.synchronize()
                self.frame = 0
                return False
            self.pose_R, self.pose_t = self.pose_R @ R, self.pose_R @ t + self.pose_t     # kinfu.cpp:256-268
        else:
            self.pose_R, self.pose_t = np.asarray(pose[0], dtype=np.float64), np.asarray(pose[1], dtype=np.float64)
        # kinfu.cpp:279-290: no integration if the camera (by the hint) did not move
        hint_R, hint_t = (np.eye(3), np.zeros(3)) if motion_hint is None else motion_hint
        rnorm = np.float32(math.acos(min(1.0, max(-1.0, (np.trace(np.float64(hint_R)) - 1.0) / 2.0))))
        tnorm = np.float32(np.linalg.norm(np.float64(hint_t)))
        if (rnorm + tnorm) / 2 >= self.min_move:
            self._stage("integrate", self._integrate)
            self.integrated = True
        self._stage("raycast", self._raycast_model)
        self.ctx.synchronize()
        if self.world > 1 and self.ctx.comm_status(self.comm)[1]:
            # more pixels to settle than the composite's list holds (include/yakhip.h): once more with room for all
            self.ctx.set_option("composite_capacity", self.rows * self.cols)
            self._raycast_model()
            self.ctx.synchronize()
            self.ctx.set_option("composite_capacity", 0)
        self.frame += 1
        return True

    def own_layers(self):
        """This rank's own layers of the volume (a view into the held tensor)."""
        plane = self.dims[0] * self.dims[1]
        a = (self.own[0] - self.held[0]) * plane
        return self.volume[a:a + (self.own[1] - self.own[0]) * plane]


def make_comm(ctx, dist):
    """The yakhip communicator of this process's rank: RCCL when torch.distributed runs over nccl (the unique id
    travels through torch.distributed once, the collectives of the frame loop are then the library's own, on its
    stream), else a caller-supplied all-reduce that stages the values through the host for gloo."""
    import torch

    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    if world == 1:
        return capi.Comm.local(1)[0]
    rank = dist.get_rank()
    if dist.get_backend() == "nccl":
        box = [capi.Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return capi.Comm.rccl(ctx, box[0], world, rank)

    def fn(buf, count, op, stream):
        host = np.empty(count, dtype=np.float32 if op == capi.SUM_F32 else np.int32)
        ctx.download(host.ctypes.data, buf, count * 4)            # blocking, ordered on the context's stream
        t = torch.from_numpy(host)
        dist.all_reduce(t, op=dist.ReduceOp.MIN if op == capi.MIN_I32 else dist.ReduceOp.SUM)
        ctx.upload(buf, host.ctypes.data, count * 4)
        return 0

    return capi.Comm.custom(world, rank, fn)
