"""The complete KinFu loop (reference yak/src/kfusion/kinfu.cpp:162-309) driven through the C ABI exactly as
yak_amd/host/kinfu.cpp drives it: compute_dists, bilateral, pyramid, point/normal maps, ICP (10/5/4
linearisations on levels 0/1/2 as projective_icp.cpp:206 indexes iters_ by level; coarse to fine),
integrate, raycast, 2x2 resize -- on the S2 orbit with 1 mm depth noise, depth frames resident on the
device.  Used by bench.py (fused frames/s, BASELINE metric part 3) and tools/bench_loop.py.  GPU only.
"""
import math
import time

import numpy as np


def rodrigues(r):
    th = np.linalg.norm(r)
    if th < 1e-12:
        return np.eye(3)
    k = r / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * K @ K


def run_kinfu_loop(ctx, n=512, frames=60, icp="device", scene=None, on_system=None, surface_map=True, overlap=True, prepare=True):
    """icp: "device" (yakhip_icp_estimate, one wait per frame), "host" (one blocking linearisation at a time,
    numpy solve) or None (true poses supplied).  scene: the frames, if the caller made them already (a list of
    synth.scene_s2(f, noise_mm=1.0) for f in range(frames)).  overlap: the frame's pre-processing goes to the context's
    second stream (yakhip_ctx_side_begin), where it runs while the integrate and ray cast of the frame before are still
    in flight; the host never waits for the device except for the ICP result.  Returns a dict with frames/s and
    per-stage stream times."""
    import torch

    from . import capi, synth

    rows, cols, intr = synth.ROWS, synth.COLS, synth.INTR
    dims, vs, tr = synth.volume_geometry(n)
    vol = torch.zeros(n ** 3, dtype=torch.int32, device="cuda")
    desc = capi.volume_desc(vol.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT)
    if surface_map:  # integrate keeps it, the ray cast skips the fetches of free space by it (include/yakhip.h)
        smap = torch.zeros(capi.surface_map_bytes(desc) // 4, dtype=torch.int32, device="cuda")
        desc = capi.volume_desc(vol.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT, surface_map=smap.data_ptr())
    ctx.clear_volume(desc)
    i16 = lambda r, c: torch.empty((r, c), dtype=torch.int16, device="cuda")        # noqa: E731
    f4 = lambda r, c: torch.empty((r, c, 4), dtype=torch.float32, device="cuda")    # noqa: E731
    LEVELS, iters = 3, (10, 5, 4)  # iterations on levels 0, 1, 2 (iters_[level_index], projective_icp.cpp:206)
    dists2 = [i16(rows, cols), i16(rows, cols)]  # two: frame f + 1 is prepared while frame f is still being integrated
    cur_d = [i16(rows >> l, cols >> l) for l in range(LEVELS)]
    cur_p = [f4(rows >> l, cols >> l) for l in range(LEVELS)]
    cur_n = [f4(rows >> l, cols >> l) for l in range(LEVELS)]
    prv_p = [f4(rows >> l, cols >> l) for l in range(LEVELS)]
    prv_n = [f4(rows >> l, cols >> l) for l in range(LEVELS)]

    n_frames = int(frames)
    if scene is None:
        scene = [synth.scene_s2(f, noise_mm=1.0) for f in range(n_frames)]
    assert len(scene) >= n_frames
    dev_depth = [torch.from_numpy(d.view(np.int16)).cuda() for d, _, _ in scene[:n_frames]]
    # poses are camera -> volume; frame 0 is integrated at its true pose
    def c2v(R_v2c, t_v2c):
        R = R_v2c.astype(np.float64).T
        return R, -R @ t_v2c.astype(np.float64)
    pose_R, pose_t = c2v(scene[0][1], scene[0][2])

    stage = {k: 0.0 for k in ("preprocess", "icp", "integrate", "raycast")}
    ev = lambda: torch.cuda.Event(enable_timing=True)   # noqa: E731
    # the collector's pauses land between two launches and show up as stream time (a 33 ms full collection at
    # frame 237 of the orbit, 1 ms ones elsewhere): none inside the timed loop
    import gc
    gc.collect()
    gc_was_enabled = gc.isenabled()
    gc.disable()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    worst = 0.0
    icp_ms = []
    events = []
    for f in range(n_frames):
        e = [ev() for _ in range(5)]
        events.append(e)
        d = dev_depth[f]
        dists = dists2[f & 1]
        if overlap:
            ctx.side_begin()
        e[0].record(torch.cuda.ExternalStream(ctx.stream_handle()) if overlap else None)
        ctx.compute_dists(d.data_ptr(), cols * 2, dists.data_ptr(), cols * 2, rows, cols, intr)
        if overlap and prepare:
            ctx.integrate_prepare(desc, dists.data_ptr(), cols * 2, rows, cols)
        ctx.bilateral_filter(d.data_ptr(), cols * 2, cur_d[0].data_ptr(), cols * 2, rows, cols, 7, 4.5, 0.04)
        for l in range(1, LEVELS):
            ctx.depth_pyramid(cur_d[l - 1].data_ptr(), (cols >> (l - 1)) * 2, rows >> (l - 1), cols >> (l - 1),
                              cur_d[l].data_ptr(), (cols >> l) * 2, 0.04)
        for l in range(LEVELS):
            li = tuple(np.float32(v) / np.float32(1 << l) for v in intr)
            ctx.compute_points_normals(li, cur_d[l].data_ptr(), (cols >> l) * 2, rows >> l, cols >> l, cur_p[l].data_ptr(),
                                       (cols >> l) * 16, cur_n[l].data_ptr(), (cols >> l) * 16)
        if overlap:
            pre_end = ev()
            pre_end.record(torch.cuda.ExternalStream(ctx.stream_handle()))
            e.append(pre_end)
            ctx.side_end()
        e[1].record()
        if f > 0 and icp is not None:
            if icp == "host":
                R, t = np.eye(3), np.zeros(3)      # current -> previous camera
                for l in reversed(range(LEVELS)):
                    r_, c_ = rows >> l, cols >> l
                    for _ in range(iters[l]):
                        sums, _ = ctx.icp_accumulate(np.float32(R).reshape(-1), np.float32(t), intr, l, 0.1,
                                                     math.radians(30.0), cur_p[l].data_ptr(), c_ * 16, cur_n[l].data_ptr(),
                                                     c_ * 16, prv_p[l].data_ptr(), c_ * 16, prv_n[l].data_ptr(), c_ * 16,
                                                     r_, c_)
                        A, b, k = np.zeros((6, 6)), np.zeros(6), 0
                        for i in range(6):
                            for j in range(i, 7):
                                if j == 6:
                                    b[i] = sums[k]
                                else:
                                    A[i, j] = A[j, i] = sums[k]
                                k += 1
                        if on_system is not None:  # diagnostics: on_system(frame, level, A, b)
                            on_system(f, l, A, b)
                        det = np.linalg.det(A)
                        if not abs(det) >= 1e-15:  # projective_icp.cpp:215-222 (NaN fails too): estimateTransform gives up
                            raise RuntimeError("ICP: singular normal equations at frame %d" % f)
                        x = np.linalg.solve(A, b)
                        Ri = rodrigues(x[:3])
                        R, t = Ri @ R, Ri @ t + x[3:]
            else:
                lv = [(cur_p[l].data_ptr(), cur_n[l].data_ptr(), prv_p[l].data_ptr(), prv_n[l].data_ptr(), (cols >> l) * 16,
                       rows >> l, cols >> l, iters[l]) for l in range(LEVELS)]
                Rl, tl, ok = ctx.icp_estimate(np.eye(3).reshape(-1), np.zeros(3), intr, 0.1, math.radians(30.0), lv)
                assert ok
                R, t = np.array(Rl, dtype=np.float64).reshape(3, 3), np.array(tl, dtype=np.float64)
            pose_R, pose_t = pose_R @ R, pose_R @ t + pose_t
        elif f > 0:
            pose_R, pose_t = c2v(scene[f][1], scene[f][2])
        e[2].record()
        Rv = pose_R.T
        tv = -Rv @ pose_t
        ctx.integrate(desc, dists.data_ptr(), cols * 2, rows, cols, np.float32(Rv).reshape(-1), np.float32(tv), intr)
        e[3].record()
        if f == 0:
            # kinfu.cpp:191-201: frame 0 integrates, its OWN point / normal maps become the first model maps
            # (the pyramids are swapped), and there is no ray cast
            cur_p, prv_p = prv_p, cur_p
            cur_n, prv_n = prv_n, cur_n
        else:
            assert LEVELS == 3
            ctx.raycast_points_pyramid(desc, np.float32(pose_R).reshape(-1), np.float32(pose_t), np.float32(Rv).reshape(-1), intr,
                                       prv_p[0].data_ptr(), cols * 16, prv_n[0].data_ptr(), cols * 16, rows, cols,
                                       (prv_p[1].data_ptr(), (cols >> 1) * 16, prv_n[1].data_ptr(), (cols >> 1) * 16),
                                       (prv_p[2].data_ptr(), (cols >> 2) * 16, prv_n[2].data_ptr(), (cols >> 2) * 16))
        e[4].record()
        tR, tt = c2v(scene[f][1], scene[f][2])
        worst = max(worst, float(np.linalg.norm(pose_t - tt)))
    torch.cuda.synchronize()
    wall = time.perf_counter() - t_start
    for e in events:
        for name, a, b_ in (("preprocess", 0, 5 if overlap else 1), ("icp", 1, 2), ("integrate", 2, 3), ("raycast", 3, 4)):
            stage[name] += e[a].elapsed_time(e[b_])
        icp_ms.append(e[1].elapsed_time(e[2]))
    if gc_was_enabled:
        gc.enable()
    checksum = int((vol.to(torch.int64) * (torch.arange(vol.numel(), device="cuda", dtype=torch.int64) % 65521 + 1)).sum().item())
    return {"n": n, "frames": n_frames, "icp": icp, "frames_per_s": n_frames / wall, "wall_s": wall,
            "final_pose": (np.float64(pose_R).tolist(), np.float64(pose_t).tolist()), "volume_checksum": checksum,
            "worst_position_error_mm": worst * 1e3, "stage_ms": {k: v / n_frames for k, v in stage.items()},
            "icp_ms_per_frame": icp_ms}
