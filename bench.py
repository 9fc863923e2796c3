#!/usr/bin/env python3
"""bench.py -- TsdfVolume::integrate throughput on MI355X (BASELINE.json: voxel updates/sec).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path over one batch of synthetic input: one 640x480 depth frame
(already converted to the half ray-distance image and resident in HBM) integrated into the rank's
z-slab of the TSDF volume.  N = 1 is BASELINE.json configs[1]: 512^3 volume, scene S0 ("wall":
every voxel is in view and in front of the surface, so every voxel is read, updated and written).
N > 1: the volume grows with N (z-slab sharding, 512^3 voxels per rank, no data-path collective:
every rank integrates its own layers from the same frame) -> weak scaling.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      HBM roofline of the integrate kernel (algorithmic bytes / mean launch time)
  cpu_baseline  the scalar CPU oracle timed on this box's host cores (N = 1, rank 0 only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)
VOXELS_PER_RANK_EDGE = 512
CEILING_SWEEPS = 100    # in-place sweeps of the slab that calibrate the box's RMW ceiling before the timed region
STEADY_LAUNCHES = 100   # launches after the timed region whose second half gives the steady-state kernel time
SERIES_LEN = 80         # per-launch kernel times printed, from the first launch of the process

from yak_amd.shard import BENCH_GLOBAL_DIMS as GLOBAL_DIMS, slab_range  # noqa: E402


def wall_scene(dims, voxel):
    """Scene S0 generalised to any volume extent: camera centred on the x/y extent, far enough back
    that the near face projects inside the image, wall 0.3 m behind the far face."""
    from yak_amd import synth

    ex, ey, ez = (d * voxel for d in dims)
    back = 3.7 * max(ex, ey) / 3.0
    eye = np.array([ex / 2, ey / 2, -back])
    R_c2v = np.eye(3)
    depth = synth.render_depth(R_c2v, eye, planes=[((0, 0, ez + 0.3), (0, 0, 1))])
    R, t = synth.vol2cam(R_c2v, eye)
    return depth, R, t


def cpu_baseline(dims, vs, tr, dists, R, t, intr, max_weight):
    """The scalar oracle (kind 'port': the reference's integrate is CUDA-only) on the host cores."""
    import oracle

    threads = min(oracle.max_threads(), os.cpu_count() or 1)
    layers = 128  # bounded sample: the first 128 of 512 layers, same frame, same pose
    n = dims[0] * dims[1] * layers
    slab = np.zeros(n, dtype=np.uint32)
    t0 = time.perf_counter()
    _, upd1 = oracle.integrate(slab, dims, vs, tr, max_weight, dists, R, t, intr, z0=0, z1=layers, nthreads=1)
    t1 = time.perf_counter() - t0
    reps, upd, tm = 0, 0, 0.0
    while tm < 10.0 and reps < 2000:  # about 10 s of all-core work (plus the one-thread pass above)
        slab[:] = 0
        t0 = time.perf_counter()
        _, u = oracle.integrate(slab, dims, vs, tr, max_weight, dists, R, t, intr, z0=0, z1=layers, nthreads=threads)
        tm += time.perf_counter() - t0
        upd += u
        reps += 1
    return {
        "value": round(upd / tm / 1e9, 4),
        "unit": "Gvox/s",
        "cores": threads,
        "kind": "port",
        "value_1thread": round(upd1 / t1 / 1e9, 4),
        "sample": "scalar C oracle (oracle/yak_oracle.c, OpenMP over y), same S0 frame, layers [0,%d) of the "
                  "%dx%dx%d volume (%d voxels), %d repetitions at %d threads + 1 at 1 thread" % (
                      layers, dims[0], dims[1], dims[2], n, reps, threads),
    }


def next_rows(ctx, torch, capi, synth, with_cpu):
    """Secondary measurements, N = 1 only: the SURVEY.md 8f rows that sit either side of the hot path, on a
    512^3 volume fused from eight views of the S2 orbit scene.  Device times from the library's HIP event
    pairs around the kernels; never part of `value`."""
    import math

    n = VOXELS_PER_RANK_EDGE
    rows, cols, intr = synth.ROWS, synth.COLS, synth.INTR
    dims, vs, tr = synth.volume_geometry(n)
    vol = torch.zeros(n ** 3, dtype=torch.int32, device="cuda")
    plain = capi.volume_desc(vol.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT)
    # with its surface map (one bit per 16^3 brick, kept by integrate, consulted by the ray cast: include/yakhip.h)
    smap = torch.zeros(capi.surface_map_bytes(plain) // 4, dtype=torch.int32, device="cuda")
    desc = capi.volume_desc(vol.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT, surface_map=smap.data_ptr())
    frames, keep = [], []
    for f in range(0, 300, 38):
        depth, R, t = synth.scene_s2(f, noise_mm=1.0)  # (the frames of the `orbit_views` row)
        d = torch.from_numpy(depth.view(np.int16)).cuda()
        dd = torch.empty_like(d)
        ctx.compute_dists(d.data_ptr(), cols * 2, dd.data_ptr(), cols * 2, rows, cols, intr)
        keep.append((d, dd))
        frames.append((dd.data_ptr(), cols * 2, rows, cols, R.reshape(-1), t, intr))

    def timed(fn, reps, wait=False):
        fn()
        if wait:
            torch.cuda.synchronize()
        ctx.set_option("kernel_timing", 1)
        for _ in range(reps):
            fn()
            if wait:  # (a caller that waits once per call: see orbit_views_row)
                torch.cuda.synchronize()
        ms, launches = ctx.kernel_time()
        ctx.set_option("kernel_timing", 0)
        return ms / reps

    out = {"volume": "512^3 fused from 8 views of the S2 orbit scene (ground plane + 3 spheres)"}
    # The volume is built with the batched kernel only, so that every integrate_kernel launch of this
    # command belongs to the headline workload (the rocprofv3 average of that kernel stays comparable).
    ctx.integrate_batch(desc, frames[:4])
    ctx.integrate_batch(desc, frames[4:])
    # (the batched entry point on the first four frames of the `orbit_views` row: where the brick test found mostly surface
    #  in the last launches, the library runs the batch as single-frame launches -- the voxels are the same by definition)
    c0 = ctx.launch_counts()
    # (like for like with `orbit_views.kernel_ms_mean`: the same eight frames, the same noise, a volume without a surface map)

    def both_batches():
        ctx.integrate_batch(plain, frames[:4])
        ctx.integrate_batch(plain, frames[4:])

    out["integrate_batch_k4_s2_ms_per_frame"] = round(timed(both_batches, 5, wait=True) / 8, 4)

    # ... and the same eight frames as single calls under the same protocol (this volume, one wait per four frames), so that
    # the two figures differ by the entry point alone
    def eight_singles():
        for i, fr in enumerate(frames):
            ctx.integrate(plain, *fr)
            if i == 3:
                torch.cuda.synchronize()

    out["integrate_singles_same_protocol_ms_per_frame"] = round(timed(eight_singles, 5, wait=True) / 8, 4)
    c1 = ctx.launch_counts()
    out["integrate_batch_k4_s2_ran_as"] = ("single-frame launches: %s" % {k: c1[k] - c0[k] for k in c1}
                                           if sum(c1.values()) > sum(c0.values()) else "integrate_batch_kernel<4> sweeps")

    # raycast from orbit poses (cam2vol = inverse of vol2cam): along the volume's z axis (views 0, 150) and
    # oblique (37, 100, 222), with the surface map and without it
    pts = torch.empty((rows, cols, 4), dtype=torch.float32, device="cuda")
    nrm = torch.empty_like(pts)
    per_view, per_view_plain, samples, hits = {}, {}, 0, 0
    step = tr * 0.75
    for view in (0, 37, 100, 150, 222):
        _, R_v2c, t_v2c = synth.scene_s2(view)
        Rc = np.ascontiguousarray(R_v2c.T)
        tc = (-(R_v2c.astype(np.float64).T @ t_v2c.astype(np.float64))).astype(np.float32)
        for d, sink in ((plain, per_view_plain), (desc, per_view)):
            sink[str(view)] = round(timed(lambda: ctx.raycast_points(d, Rc.reshape(-1), tc, R_v2c.reshape(-1), intr, pts.data_ptr(),
                                                                     cols * 16, nrm.data_ptr(), cols * 16, rows, cols), 20), 4)
        # samples the march takes (tsdf_volume.cu:149-165, 306-324): from the volume's near face to the hit, or
        # to its far face; float64 on the host, a count for the byte figure below, not a parity statement
        v = pts.cpu().numpy().astype(np.float64)
        hit = ~np.isnan(v[..., 0])
        u, w = np.meshgrid((np.arange(cols) - intr[2]) / intr[0], (np.arange(rows) - intr[3]) / intr[1])
        dirs = np.stack([u, w, np.ones_like(u)], -1) @ Rc.astype(np.float64).T
        dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
        with np.errstate(divide="ignore", invalid="ignore"):
            box = np.array([vs[i] * dims[i] - vs[i] for i in range(3)])
            ta, tb = (0.0 - tc) / dirs, (box - tc) / dirs
        tmin = np.maximum(np.minimum(ta, tb).max(-1), 0.0)
        tmax = np.maximum(ta, tb).min(-1)
        t_end = np.where(hit, np.linalg.norm(v[..., :3], axis=-1), tmax)
        samples += int(np.clip((t_end - tmin) / step, 0, None)[tmin < tmax].sum())
        hits += int(hit.sum())
    ms = float(np.mean(list(per_view.values())))
    # per ray cast: one half per sample, 2 + 6 trilinear taps of 8 halves per hit, 32 B of maps per pixel
    rc_bytes = (2 * samples + 2 * 64 * hits) // 5 + 32 * rows * cols
    out["raycast"] = {"ms": round(ms, 4), "ms_per_view": per_view, "ms_per_view_without_surface_map": per_view_plain,
                      "rays": rows * cols, "hits_mean": hits // 5, "samples_mean": samples // 5,
                      "roofline": {"bound": "hbm", "achieved": round(rc_bytes / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS,
                                   "unit": "GB/s", "frac": round(rc_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                   "algorithmic_bytes": rc_bytes, "traffic": load_traffic("raycast_s2_n1")},
                      "note": "not bound by HBM: the march is one dependent chain of samples per ray (position, nearest voxel, "
                              "sign test); with the surface map the samples of free space are settled from LDS and the kernel "
                              "is bound by VALU issue, without it by gathers of 2 B per 64 B sector"}

    # one ICP linearisation at full resolution: the ray-cast maps against themselves
    def icp():
        ctx.icp_accumulate(np.eye(3).reshape(-1), np.zeros(3), intr, 0, 0.1, math.radians(30.0), pts.data_ptr(), cols * 16,
                           nrm.data_ptr(), cols * 16, pts.data_ptr(), cols * 16, nrm.data_ptr(), cols * 16, rows, cols)
    icp()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        icp()
    out["icp_accumulate_level0_ms_blocking"] = round((time.perf_counter() - t0) / 20 * 1e3, 4)

    # marching cubes: count pass + scan + emit pass; HBM roofline on 2 reads of the volume + the triangles
    ntri = ctx.marching_cubes(plain, 2, vs[0])
    tri = torch.empty((max(ntri, 1), 9), dtype=torch.float32, device="cuda")
    ms = timed(lambda: ctx.marching_cubes(plain, 2, vs[0], tri.data_ptr(), ntri), 5)
    # with the volume's surface map the count pass reads only near marked bricks (the same triangles)
    ms_map = timed(lambda: ctx.marching_cubes(desc, 2, vs[0], tri.data_ptr(), ntri), 5)
    # the two passes on their own (the library brackets count + scans + list, and the emit pass, with an event pair each)
    ctx.set_option("kernel_timing", 1)
    for _ in range(5):
        ctx.marching_cubes(plain, 2, vs[0], tri.data_ptr(), ntri)
    pass_ms, _ = ctx.kernel_times(10)
    ctx.set_option("kernel_timing", 0)
    # algorithmic bytes (DESIGN 3.8 uses the same definition): ONE read of the volume + the triangles written -- what an
    # extraction without prior knowledge must move; the emit pass's second look at the tiles that hold surface is the
    # implementation's own cost
    mc_bytes = 4 * n ** 3 + 36 * ntri
    out["marching_cubes"] = {"ms_both_passes": round(ms, 4), "ms_both_passes_with_surface_map": round(ms_map, 4),
                             "ms_count_scan_list": round(sum(pass_ms[0::2]) / 5, 4), "ms_emit": round(sum(pass_ms[1::2]) / 5, 4),
                             "triangles": ntri, "algorithmic_bytes": mc_bytes,
                             "achieved_GBps": round(mc_bytes / (ms * 1e-3) / 1e9, 1),
                             "frac_of_hbm_peak": round(mc_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                             "Gvox_per_s": round(n ** 3 / (ms * 1e-3) / 1e9, 1)}
    # vertex welding of that triangle list (MarchingCubesParameters::clean): wall time of the blocking call (its scratch
    # is kept in the context: the first call above allocated it)
    wv = torch.empty((max(3 * ntri, 1), 3), dtype=torch.float32, device="cuda")
    wi = torch.empty((max(3 * ntri, 1),), dtype=torch.int32, device="cuda")
    nvert = ctx.weld_vertices(tri.data_ptr(), ntri, wv.data_ptr(), wi.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.weld_vertices(tri.data_ptr(), ntri, wv.data_ptr(), wi.data_ptr())
    out["marching_cubes"]["weld_vertices"] = {"ms_blocking": round((time.perf_counter() - t0) / 5 * 1e3, 4),
                                              "corners": 3 * ntri, "vertices": nvert}
    del wv, wi
    # cloud extraction (count, scan, list, emit over the tiles that hold points): ONE read of the volume + 16 B per point,
    # the same definition as marching cubes' (two reads were counted here until the emit pass stopped re-reading the volume)
    npts = ctx.extract_cloud(plain, np.eye(3).reshape(-1), np.zeros(3))
    cloud = torch.empty((max(npts, 1), 4), dtype=torch.float32, device="cuda")
    ms = timed(lambda: ctx.extract_cloud(plain, np.eye(3).reshape(-1), np.zeros(3), cloud.data_ptr(), npts), 5)
    ms_map = timed(lambda: ctx.extract_cloud(desc, np.eye(3).reshape(-1), np.zeros(3), cloud.data_ptr(), npts), 5)
    cl_bytes = 4 * n ** 3 + 16 * npts
    out["extract_cloud"] = {"ms_both_passes": round(ms, 4), "ms_both_passes_with_surface_map": round(ms_map, 4), "points": npts,
                            "algorithmic_bytes": cl_bytes,
                            "achieved_GBps": round(cl_bytes / (ms * 1e-3) / 1e9, 1),
                            "frac_of_hbm_peak": round(cl_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
    if with_cpu:
        out["marching_cubes"]["cpu_baseline"] = c1_cpu_baseline(synth)
        # the same C1 volume on the GPU, for a like-for-like figure
        c1 = torch.from_numpy(synth.sphere_tsdf_volume(256, 25.6).view(np.int32)).cuda()
        c1desc = capi.volume_desc(c1.data_ptr(), (256, 256, 256), (1.0, 1.0, 1.0), 4.0, synth.MAX_WEIGHT)
        n1 = ctx.marching_cubes(c1desc, 4, 1.0)
        t1 = torch.empty((max(n1, 1), 9), dtype=torch.float32, device="cuda")
        ms1 = timed(lambda: ctx.marching_cubes(c1desc, 4, 1.0, t1.data_ptr(), n1), 5)
        out["marching_cubes"]["c1_sphere_256"] = {"ms_both_passes": round(ms1, 4), "triangles": n1,
                                                  "Gvox_per_s": round(256 ** 3 / (ms1 * 1e-3) / 1e9, 1)}
    return out


def kinfu_loop_row(ctx):
    """BASELINE metric part 3: fused frames per second of the full loop (integrate + raycast + projective ICP)
    at 512^3 on the S2 orbit, tracking from depth alone; and with supplied poses."""
    from yak_amd.loop import run_kinfu_loop

    from yak_amd import synth

    frames = 300  # BASELINE.json configs[2]: the whole orbit
    scene = [synth.scene_s2(f, noise_mm=1.0) for f in range(frames)]
    run_kinfu_loop(ctx, VOXELS_PER_RANK_EDGE, 8, "device", scene)  # warm-up: allocations, first launches
    # wall-clock rates of a 0.25 s run: the better of two runs each (the host side is Python and the box is shared)
    best = lambda a, b: a if a["frames_per_s"] >= b["frames_per_s"] else b  # noqa: E731
    r = best(run_kinfu_loop(ctx, VOXELS_PER_RANK_EDGE, frames, "device", scene),
             run_kinfu_loop(ctx, VOXELS_PER_RANK_EDGE, frames, "device", scene))
    p = best(run_kinfu_loop(ctx, VOXELS_PER_RANK_EDGE, frames, None, scene),
             run_kinfu_loop(ctx, VOXELS_PER_RANK_EDGE, frames, None, scene))
    out = {"workload": "512^3, the 300 frames of the S2 orbit (1.2 deg per frame, 1 mm depth noise), depth frames resident in HBM; "
                       "per frame: compute_dists, bilateral, 3-level pyramid, point/normal maps, ICP 10/5/4 linearisations "
                       "(device-resident loop), integrate, raycast with the maps' 2x2 pyramid; host waits once per frame",
           "host": "Python over the C ABI (yak_amd/loop.py)",
           "fused_frames_per_s": round(r["frames_per_s"], 1),
           "fused_frames_per_s_supplied_poses": round(p["frames_per_s"], 1),
           "worst_position_error_mm": round(r["worst_position_error_mm"], 2),
           "stage_ms": {k: round(v, 4) for k, v in r["stage_ms"].items()}}
    cpp = cpp_loop_rates(frames)
    if cpp:
        out["cpp_host"] = cpp
    return out


SHARDED_LOOP_EDGE = 1024   # BASELINE config 4: the 1024^3 volume of the S2 orbit, z-slabs over the ranks
SHARDED_LOOP_FRAMES = 60


def kinfu_loop_sharded_row(ctx, torch, capi, synth, dist, world, rank, frames=SHARDED_LOOP_FRAMES, edge=SHARDED_LOOP_EDGE):
    """BASELINE config 4: the tracking loop over a z-slab sharded volume, one rank per GPU (yak_amd/kinfu_sharded.py over
    yakhip_icp_estimate_sharded / yakhip_raycast_sharded: 19 all-reduces of 33 floats, the ray cast's composite -- MIN
    over the keys, SUM over the winners' payload, the settled pixels once more).  Fused frames per second of the
    slowest rank, stream time per stage (maximum over the ranks), and the frame's collectives once more on their own:
    the same buffers and counts, nothing else on the stream.  All ranks call this; rank 0 gets the row."""
    from yak_amd.kinfu_sharded import ShardedKinFu

    rows, cols, intr = synth.ROWS, synth.COLS, synth.INTR
    dims, vs, tr = synth.volume_geometry(edge)
    kf = ShardedKinFu(ctx, dist if world > 1 else None, dims, vs, tr, synth.MAX_WEIGHT, intr, rows, cols)
    scene = [synth.scene_s2(f, noise_mm=1.0) for f in range(frames)]
    depth = [torch.from_numpy(d.view(np.int16)).cuda() for d, _, _ in scene]

    def c2v(R_v2c, t_v2c):
        R = R_v2c.astype(np.float64).T
        return R, -R @ t_v2c.astype(np.float64)

    def run(collect):
        kf.ctx.clear_volume(kf.desc)
        kf.frame = 0
        kf.stage_events = [] if collect else None
        fused = 0
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for f in range(frames):
            fused += bool(kf(depth[f], first_pose=c2v(scene[0][1], scene[0][2]) if f == 0 else None))
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        return time.perf_counter() - t0, fused

    run(False)  # warm-up: allocations, first launches, the communicator's first collectives
    elapsed, fused = run(True)
    stage = {}
    for name, e0, e1 in kf.stage_events:
        stage.setdefault(name, []).append(e0.elapsed_time(e1))
    stage_ms = {k: sum(v) / len(v) for k, v in stage.items()}
    R_true, t_true = c2v(scene[frames - 1][1], scene[frames - 1][2])
    err_mm = float(np.linalg.norm(kf.pose_t - t_true) * 1e3)

    # the collectives of one frame on their own (what DESIGN.md 5 tabulates): 19 x 33 floats SUM, the keys MIN, the
    # payload SUM (4 words per pixel: march time + normal), the settled pixels' list SUM (16 B x max(4096, pixels / 64) slots)
    small = torch.zeros(64, dtype=torch.float32, device="cuda")
    keys = torch.zeros(rows * cols, dtype=torch.int32, device="cuda")
    payload = torch.zeros(rows * cols * 4, dtype=torch.int32, device="cuda")
    slots = torch.zeros(4 * max(4096, rows * cols // 64), dtype=torch.int32, device="cuda")

    def timed_coll(fn, reps=10):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    def icp_sums():
        for _ in range(19):
            ctx.allreduce(kf.comm, small.data_ptr(), 33, capi.SUM_F32)

    coll = {"icp_19_x_33_floats_ms": timed_coll(icp_sums),
            "raycast_keys_min_ms": timed_coll(lambda: ctx.allreduce(kf.comm, keys.data_ptr(), rows * cols, capi.MIN_I32)),
            "raycast_payload_sum_ms": timed_coll(lambda: ctx.allreduce(kf.comm, payload.data_ptr(), rows * cols * 4, capi.SUM_I32)),
            "raycast_settled_sum_ms": timed_coll(lambda: ctx.allreduce(kf.comm, slots.data_ptr(), slots.numel(), capi.SUM_I32))}
    vals = [elapsed, err_mm] + [stage_ms.get(k, 0.0) for k in ("preprocess", "icp", "integrate", "raycast")] + list(coll.values())
    t = torch.tensor(vals, dtype=torch.float64, device="cpu" if dist is not None and world > 1 and dist.get_backend() == "gloo" else "cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    vals = [float(v) for v in t]
    transport = "one rank: no exchange" if world == 1 else ("RCCL (ncclAllReduce on the context's stream)" if dist.get_backend() == "nccl"
                                                                else "gloo through the host (a rehearsal: not the transport of a real run)")
    out = {"workload": "BASELINE config 4: %d^3 volume in %d z-slabs (+ the ray cast's halo, integrated redundantly), the first %d frames "
                       "of the S2 orbit (1 mm depth noise) tracked from depth alone; per frame: pre-processing (replicated), ICP over "
                       "row bands with 19 all-reduces of 33 floats, integrate (no exchange), slab ray cast + composite"
                       % (edge, world, frames),
           "host": "Python over the C ABI (yak_amd/kinfu_sharded.py), one process per GPU", "transport": transport,
           "frames": frames, "frames_fused": fused, "fused_frames_per_s": round(frames / vals[0], 1),
           "position_error_last_frame_mm": round(vals[1], 2),
           "stage_ms_slowest_rank": {k: round(v, 4) for k, v in zip(("preprocess", "icp", "integrate", "raycast"), vals[2:6])},
           "collectives_of_a_frame_on_their_own_ms": {k: round(v, 4) for k, v in zip(coll.keys(), vals[6:])},
           "held_layers_rank0": list(kf.held), "halo_layers": kf.halo}
    out["collectives_of_a_frame_on_their_own_ms"]["sum"] = round(sum(vals[6:]), 4)
    del kf, depth
    torch.cuda.empty_cache()
    return out if rank == 0 else None


def cpp_loop_rates(frames):
    """The same orbit through the C++ classes (kfusion::KinFu in examples/offline_fuse, a child process): its frames
    come from host memory, so every frame's upload is inside the figure.  None if the example is not built."""
    import re
    import subprocess

    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "examples", "offline_fuse")
    if not os.access(exe, os.X_OK):
        return None
    out = {"what": "examples/offline_fuse --n 512 --frames %d [--track]: kfusion::KinFu over the C ABI, per-frame upload included" % frames}
    for key, extra in (("fused_frames_per_s", ["--track"]), ("fused_frames_per_s_supplied_poses", [])):
        try:
            txt = subprocess.run([exe, "--n", "512", "--frames", str(frames)] + extra, capture_output=True, text=True, timeout=120).stdout
        except (OSError, subprocess.TimeoutExpired):
            return None
        m = re.search(r"= (\d+) frames/s", txt)
        if not m:
            return None
        out[key] = int(m.group(1))
        m = re.search(r"\((\d+) frames/s after the first (\d+)\)", txt)
        if m:  # without the frames that carry allocations and first launches
            out[key + "_after_the_first_%s" % m.group(2)] = int(m.group(1))
    return out


def orbit_views_row(ctx, torch, capi, synth):
    """The hot path on surface-rich frames (the S0 wall is the kernel's best case): single-frame
    integrate of 8 views of the S2 orbit into one 512^3 volume, kernel time per view from the library's event
    pairs, voxels updated per view counted from the weights."""
    n = VOXELS_PER_RANK_EDGE
    rows, cols, intr = synth.ROWS, synth.COLS, synth.INTR
    dims, vs, tr = synth.volume_geometry(n)
    vol = torch.zeros(n ** 3, dtype=torch.int32, device="cuda")
    desc = capi.volume_desc(vol.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT)
    views, keep = [], []
    for f in range(0, 300, 38):
        depth, R, t = synth.scene_s2(f, noise_mm=1.0)
        d = torch.from_numpy(depth.view(np.int16)).cuda()
        dd = torch.empty_like(d)
        ctx.compute_dists(d.data_ptr(), cols * 2, dd.data_ptr(), cols * 2, rows, cols, intr)
        keep.append((d, dd))
        views.append((dd.data_ptr(), R.reshape(-1), t))
    # voxels each view updates: weight sum before / after one pass from a cleared volume (weights only grow)
    updated = []
    for ptr, R, t in views:
        before = int((vol >> 16).sum(dtype=torch.int64).item())
        ctx.integrate(desc, ptr, cols * 2, rows, cols, R, t, intr)
        updated.append(int((vol >> 16).sum(dtype=torch.int64).item()) - before)
    # brick classes in a pass of their own: the counters are contended atomics and would be in the timing
    ctx.set_option("brick_stats", 1)
    for ptr, R, t in views:
        ctx.integrate(desc, ptr, cols * 2, rows, cols, R, t, intr)
    bricks = ctx.brick_stats()
    ctx.set_option("brick_stats", 0)
    # (a wait per frame, as a tracking loop has one: the library chooses between its one-kernel and its two-kernel launch
    #  by what the brick test found in the last few frames, and a caller that never waits is given one kernel)
    for ptr, R, t in views:
        ctx.integrate(desc, ptr, cols * 2, rows, cols, R, t, intr)
        torch.cuda.synchronize()
    counts0 = ctx.launch_counts()
    ctx.set_option("kernel_timing", 1)
    reps = 5
    for _ in range(reps):
        for ptr, R, t in views:
            ctx.integrate(desc, ptr, cols * 2, rows, cols, R, t, intr)
            torch.cuda.synchronize()
    ms, _ = ctx.kernel_times(reps * len(views))
    ctx.set_option("kernel_timing", 0)
    counts1 = ctx.launch_counts()
    per_view = [sum(ms[i::len(views)]) / reps for i in range(len(views))]
    mean_ms = sum(per_view) / len(per_view)
    mean_upd = sum(updated) / len(updated)
    alg = 8 * mean_upd + 2 * rows * cols
    total = max(sum(bricks.values()), 1)
    return {"workload": "512^3, 8 views of the S2 orbit (ground plane + 3 spheres, 1 mm noise), one frame per launch",
            "kernel_ms_mean": round(mean_ms, 4), "kernel_ms_per_view": [round(v, 4) for v in per_view],
            "voxels_updated_mean": int(mean_upd), "Gvox_updated_per_s": round(mean_upd / (mean_ms * 1e-3) / 1e9, 1),
            "Gvox_visited_per_s": round(n ** 3 / (mean_ms * 1e-3) / 1e9, 1),
            "algorithmic_GBps": round(alg / (mean_ms * 1e-3) / 1e9, 1),
            "frac_of_hbm_peak": round(alg / (mean_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
            "wave_bricks": {k: round(v / total, 4) for k, v in bricks.items()},
            "launches": {k: counts1[k] - counts0[k] for k in counts1},
            "kernels": "yakhip::classify_kernel<4> + yakhip::integrate_list_kernel<4, true> (two kernels: one event pair around "
                       "both) or yakhip::integrate_kernel<4, true>, see `launches`",
            "valu_instructions": load_record("integrate_orbit_views_valu", ("round2_kernel", "round3_two_kernels",
                                                                            "round3_one_kernel", "source")),
            "valu_instructions_per_updated_voxel": (round(load_record("integrate_orbit_views_valu", ("round3_two_kernels",))
                                                          ["round3_two_kernels"] * 64 / mean_upd, 1)  # (wave instructions x 64 lanes)
                                                    if load_record("integrate_orbit_views_valu", ("round3_two_kernels",)) else None),
            "note": "bound by the per-voxel path of the bricks that hold surface, not by HBM: only updated voxels "
                    "count as algorithmic bytes (SURVEY.md 8d)"}


# ---- config 5 of BASELINE.json: 2048^3 over 8 GPUs, multi-camera batched fusion + GPU marching cubes ------------
C5_EDGE = 2048           # the global volume of north_star's scaling target
C5_LAYERS_PER_RANK = 256  # 2048 / 8: every rank holds 2048 x 2048 x 256 voxels (4 GiB) at every N -> weak scaling
C5_CAMERAS = (0, 75, 150, 225)  # four views of the S2 orbit rig, 90 degrees apart, folded in one sweep


def c5_slab(world, rank):
    """Layers of rank `rank`: the N ranks hold the central 256 * N layers of the 2048-layer volume (N = 8: all of it)."""
    lo = C5_EDGE // 2 - C5_LAYERS_PER_RANK * world // 2
    return lo + C5_LAYERS_PER_RANK * rank, lo + C5_LAYERS_PER_RANK * (rank + 1)


def c1_cpu_baseline(synth):
    """Config C1: Yak's CPU marching cubes (its restatement: the reference's needs PCL/VTK) over the
    256^3 analytic sphere of the reference's stand-alone program, at 1 and at all threads."""
    import oracle

    n = 256
    vol = synth.sphere_tsdf_volume(n, 25.6)
    threads = min(oracle.max_threads(), os.cpu_count() or 1)
    t0 = time.perf_counter()
    tris = oracle.marching_cubes(vol, (n, n, n), 4, 1.0)
    t1 = time.perf_counter() - t0
    reps, tm = 0, 0.0
    while tm < 4.0 and reps < 200:
        t0 = time.perf_counter()
        oracle.marching_cubes(vol, (n, n, n), 4, 1.0, nthreads=threads)
        tm += time.perf_counter() - t0
        reps += 1
    return {"value": round(n ** 3 / (tm / reps) / 1e9, 4), "unit": "Gvox/s", "cores": threads, "kind": "port",
            "value_1thread": round(n ** 3 / t1 / 1e9, 4), "triangles": int(len(tris)),
            "sample": "oracle restatement of marchingCubesCPU (count + emit passes, x slices in parallel as the "
                      "reference's OpenMP loop) over the 256^3 sphere of radius 25.6 voxels (BASELINE config 1), "
                      "%d repetitions at %d threads + 1 at 1 thread" % (reps, threads)}


def run_c5(args, torch, capi, synth, dist, world, rank, local_rank, share_gpu, emulate):
    out = c5_measure(args, torch, capi, synth, dist, world, rank, local_rank, share_gpu, emulate)
    if out is not None:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def c5_measure(args, torch, capi, synth, dist, world, rank, local_rank, share_gpu, emulate, steps=None, warmup=None,
               with_cpu=None):
    """One step = ONE SWEEP of the rank's 2048 x 2048 x 256 slab folding the K = 4 camera frames
    (yakhip_integrate_batch); marching cubes over the slab is timed beside it (it runs once per scan, not per
    sweep).  value = voxel updates of all ranks and all K frames per second of the slowest rank."""
    n = C5_EDGE
    dims, vs, tr = synth.volume_geometry(n)
    z0, z1 = c5_slab(world, rank)
    z1h = min(z1 + 1, n)  # one layer of overlap for marching cubes (include/yakhip.h); integrated redundantly
    layers = z1h - z0
    rows, cols, intr = synth.ROWS, synth.COLS, synth.INTR
    ctx = capi.Context(local_rank)
    if args.z_chunk:
        ctx.set_option("z_chunk", args.z_chunk)
    if args.no_brick:
        ctx.set_option("brick_test", 0)
    vol = torch.empty(layers * n * n, dtype=torch.int32, device="cuda")
    held_plain = capi.volume_desc(vol.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT, z0, z1h)
    smap = torch.zeros(capi.surface_map_bytes(held_plain) // 4, dtype=torch.int32, device="cuda")
    held = capi.volume_desc(vol.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT, z0, z1h, surface_map=smap.data_ptr())
    frames, keep = [], []
    for f in C5_CAMERAS:
        depth, R, t = synth.scene_s2(f)
        d = torch.from_numpy(depth.view(np.int16)).cuda()
        dd = torch.empty_like(d)
        ctx.compute_dists(d.data_ptr(), cols * 2, dd.data_ptr(), cols * 2, rows, cols, intr)
        keep.append((d, dd, depth, R, t))
        frames.append((dd.data_ptr(), cols * 2, rows, cols, R.reshape(-1), t, intr))
    K = len(frames)
    n_steps = args.steps if steps is None else steps
    n_warm = args.warmup if warmup is None else warmup
    with_cpu = (world == 1 and not args.no_cpu_baseline) if with_cpu is None else with_cpu

    def step():
        # one wait per sweep, as a rig that hands over K frames at a time has: the library chooses between one sweep
        # of K frames and K single-frame launches by what the brick test found in the last few launches (the voxels
        # are the same), and a caller that never waits is given the sweep
        ctx.integrate_batch(held, frames)
        torch.cuda.synchronize()

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    ctx.set_option("kernel_timing", 1)
    ctx.clear_volume(held)
    step()
    torch.cuda.synchronize()
    own = vol[: (z1 - z0) * n * n]  # the overlap layer belongs to the next rank: not counted
    w = own >> 16
    touched = int((w != 0).sum().item())
    updates = int(w.sum(dtype=torch.int64).item())  # every update adds 1 to a weight (K < max_weight)
    touched_held = int(((vol >> 16) != 0).sum().item())
    for _ in range(n_warm):
        step()
    sync()
    ctx.kernel_time()  # (restarts the accumulation: what follows are the timed sweeps' launches only)
    counts0 = ctx.launch_counts()
    t0 = time.perf_counter()
    for _ in range(n_steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    kernel_total_ms, kernel_launches = ctx.kernel_time()
    counts1 = ctx.launch_counts()
    ctx.set_option("kernel_timing", 0)
    kernel_ms = kernel_total_ms / max(n_steps, 1)  # per sweep, whatever launches it was made of
    single_launches = sum(counts1.values()) - sum(counts0.values())
    sweep_kernel = ("yakhip::integrate_batch_kernel<4, true>" if single_launches == 0 else
                    "%d x (yakhip::classify_kernel<4> + yakhip::integrate_list_kernel<4, true>)" % K
                    if counts1["two_kernels"] - counts0["two_kernels"] == K * n_steps else "mixed: see launches_per_sweep")

    # marching cubes over the slab (count, scan, emit), blocking calls
    ntri = ctx.marching_cubes(held, 2, vs[0])
    tri = torch.empty((max(ntri, 1), 9), dtype=torch.float32, device="cuda")
    ctx.marching_cubes(held, 2, vs[0], tri.data_ptr(), ntri)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mc_reps = 3
    for _ in range(mc_reps):
        ctx.marching_cubes(held, 2, vs[0], tri.data_ptr(), ntri)
    mc_ms = (time.perf_counter() - t0) / mc_reps * 1e3
    t0 = time.perf_counter()
    for _ in range(mc_reps):
        ctx.marching_cubes(held_plain, 2, vs[0], tri.data_ptr(), ntri)
    mc_plain_ms = (time.perf_counter() - t0) / mc_reps * 1e3
    del tri

    # (mc_ms: with the slab's surface map, kept by the sweeps above; mc_plain_ms: without)
    totals = torch.tensor([elapsed, kernel_ms, mc_ms, float(updates), float(touched), float(ntri), mc_plain_ms], dtype=torch.float64,
                          device="cpu" if share_gpu else "cuda")
    if dist is not None:
        mx, sm = totals.clone(), totals.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed, kernel_ms, mc_ms, mc_plain_ms = float(mx[0]), float(mx[1]), float(mx[2]), float(mx[6])
        updates_total, touched_total, ntri_total = int(sm[3]), int(sm[4]), int(sm[5])
    else:
        updates_total, touched_total, ntri_total = updates, touched, ntri
    if rank == 0 or emulate:
        # algorithmic bytes of ONE launch on ONE GPU (batched mode): the touched voxels are read and
        # written once for all K frames, plus one pass over each of the K half images
        alg_bytes = 8 * touched_held + K * 2 * rows * cols
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        visited = layers * n * n
        mc_bytes = 4 * visited + 36 * ntri  # the least any extraction moves: one read of the slab + the triangles
        out = {
            "metric": "voxel updates/sec",
            "value": round(updates_total * n_steps / elapsed / 1e9, 3),
            "unit": "Gvox/s",
            "n_gpus": world,
            "steps": n_steps,
            "warmup": n_warm,
            "ms_per_step": round(elapsed / n_steps * 1e3, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE config 5: %dx%dx%d of a %d^3 volume (2048x2048x256 voxels = 4 GiB per GPU, z-slabs, "
                                   "voxel %.2f mm), one sweep per step folds %d synthetic %dx%d depth frames (orbit rig, 90 "
                                   "degrees apart, yakhip_integrate_batch); GPU marching cubes over the slab timed beside it"
                                   % (n, n, C5_LAYERS_PER_RANK * world, n, vs[0] * 1e3, K, cols, rows),
                       "frames_per_step": K, "voxel_updates_per_step": updates_total, "voxels_touched_per_step": touched_total,
                       "voxels_visited_per_step": K * C5_LAYERS_PER_RANK * world * n * n, "voxel_bytes": 4,
                       "visited_Gvox_per_s": round(K * C5_LAYERS_PER_RANK * world * n * n * n_steps / elapsed / 1e9, 1),
                       "parallelism": "z-slab x%d" % world},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         # (HBM bytes per sweep from committed counter passes of the launch kind that ran: the one sweep
                         # kernel, or K single-frame launches of two kernels each -- which touch a voxel once per frame)
                         "traffic": load_traffic(("integrate_batch_c5_rank%d_of_%d" if single_launches == 0 else
                                                  "integrate_singles_c5_rank%d_of_%d") % (rank, world)),
                         "kernel": sweep_kernel, "kernel_ms": round(kernel_ms, 5),
                         "launches_per_sweep": {"bracketed_kernel_launches": round(kernel_launches / max(n_steps, 1), 2),
                                                "single_frame_calls_as_one_kernel": (counts1["one_kernel"] - counts0["one_kernel"]) / max(n_steps, 1),
                                                "single_frame_calls_as_two_kernels": (counts1["two_kernels"] - counts0["two_kernels"]) / max(n_steps, 1)},
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "rank %d's slab, kernel time per sweep of %d frames; bound by the per-voxel path of the bricks "
                                 "that hold surface, not by HBM (only touched voxels count as algorithmic bytes, once per "
                                 "sweep)" % (rank, K)},
            "marching_cubes": {"ms_per_extraction_slowest_rank": round(mc_ms, 4),
                               "ms_per_extraction_without_surface_map": round(mc_plain_ms, 4), "triangles": ntri_total,
                               "Gvox_per_s_per_gpu": round(visited / (mc_ms * 1e-3) / 1e9, 1),
                               "algorithmic_GBps_rank0": round(mc_bytes / (mc_plain_ms * 1e-3) / 1e9, 1),
                               "frac_of_hbm_peak_rank0": round(mc_bytes / (mc_plain_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                               "how": "count pass, scan, emit pass over the slab + 1 overlap layer; wall time of the "
                                      "blocking call (the host reads the count back).  With the slab's surface map (kept by "
                                      "the integrate sweeps) the count pass reads only near marked bricks; the bytes figure "
                                      "is for the extraction WITHOUT it: one read of the slab + 36 B per triangle (the emit "
                                      "pass re-reads only tiles that hold surface)"},
        }
        if emulate:
            out["config"]["emulated_rank"] = "%d of %d (this GPU's slab only: value is per GPU)" % (rank, world)
            out["n_gpus"] = 1
        if with_cpu:
            import oracle

            threads = min(oracle.max_threads(), os.cpu_count() or 1)
            # bounded sample: the K frames applied in order to the slab's first 2 and first 10 layers.  A slab that
            # starts at layer z0 costs the port z0 replayed additions per column before its first voxel (as it costs
            # the GPU kernel's chunk starts), so T(layers) = a + b * layers: the two samples give a and b, and the
            # figure reported is for the whole slab, T(256), against the updates the GPU counted for it.
            dists_host = [dd.cpu().numpy().view(np.uint16) for d, dd, depth, R, t in keep]

            def cpu_pass(k_layers, budget_s):
                slab = np.zeros(n * n * k_layers, dtype=np.uint32)
                reps, tm = 0, 0.0
                while tm < budget_s and reps < 50:
                    slab[:] = 0
                    t0 = time.perf_counter()
                    for dh, (d, dd, depth, R, t) in zip(dists_host, keep):
                        oracle.integrate(slab, dims, vs, tr, synth.MAX_WEIGHT, dh, R, t, intr, z0=z0, z1=z0 + k_layers,
                                         nthreads=threads)
                    tm += time.perf_counter() - t0
                    reps += 1
                return tm / reps, reps

            t2, r2 = cpu_pass(2, 4.0)
            t10, r10 = cpu_pass(10, 6.0)
            b = max((t10 - t2) / 8.0, 0.0)
            t_slab = t2 + b * (C5_LAYERS_PER_RANK - 2)
            out["cpu_baseline"] = {
                "value": round(updates / t_slab / 1e9, 4), "unit": "Gvox/s", "cores": threads, "kind": "port",
                "visited_Gvox_per_s": round(K * n * n * C5_LAYERS_PER_RANK / t_slab / 1e9, 4),
                "sample": "scalar C oracle (OpenMP over y), the %d frames applied in order to layers [%d, %d) (%.3f s, %d "
                          "repetitions) and [%d, %d) (%.3f s, %d repetitions) of the slab at %d threads; extrapolated "
                          "linearly to the slab's %d layers (%.2f s per sweep)" % (
                              K, z0, z0 + 2, t2, r2, z0, z0 + 10, t10, r10, threads, C5_LAYERS_PER_RANK, t_slab)}
            out["marching_cubes"]["cpu_baseline"] = c1_cpu_baseline(synth)
    else:
        out = None
    ctx.close()
    del vol, smap
    torch.cuda.empty_cache()
    return out


def load_traffic(workload_key):
    """HBM bytes per launch from committed rocprofv3 PMC passes (profiles/traffic.json), or None."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f)
        ent = rec.get(workload_key)
        return ent.get("hbm_bytes_per_launch") if ent else None
    except (OSError, ValueError):
        return None


def load_record(key, fields):
    """Selected fields of an entry of profiles/traffic.json (committed counter passes), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            ent = json.load(f).get(key)
        return {k: ent[k] for k in fields if k in ent} if ent else None
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--scene", default="s0", choices=["s0", "s1", "s2"],
                    help="s1 = sphere+wall, s2 = a view of the orbit scene (partial updates); N=1 only")
    ap.add_argument("--config", default=os.environ.get("YAK_BENCH_CONFIG", "c2"), choices=["c2", "c5"],
                    help="c2 (default): BASELINE config 2, 512^3 per GPU, one frame per step.  c5: BASELINE config 5, "
                         "north_star's scaling target: 2048x2048x256 per GPU (2048^3 on 8), 4 cameras per sweep + GPU "
                         "marching cubes (also through the environment: YAK_BENCH_CONFIG=c5)")
    ap.add_argument("--z-chunk", type=int, default=0)
    ap.add_argument("--vx", type=int, default=0, help="cap on voxels per lane (tuning)")
    ap.add_argument("--tile", type=int, default=0, help="wave footprint: log2 lanes along x, 3..6 (tuning)")
    ap.add_argument("--zu", type=int, default=0, help="layers per loop trip (tuning)")
    ap.add_argument("--stacked", action="store_true", help="waves of a workgroup stacked in y instead of side by side (tuning)")
    ap.add_argument("--order", type=int, default=None, help="chunk_order option value (tuning)")
    ap.add_argument("--chunk-min", type=int, default=None, help="chunk_min option value (tuning)")
    ap.add_argument("--chunk-ramp", type=int, default=None, help="chunk_ramp option value (tuning)")
    ap.add_argument("--no-brick", action="store_true", help="switch the whole-brick test off (tuning)")
    ap.add_argument("--no-nt", action="store_true", help="plain instead of non-temporal volume accesses (tuning)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rows", action="store_true", help="skip the secondary measurements (raycast, ICP, MC, batch)")
    ap.add_argument("--as-rank", default="", metavar="W:R", help="diagnostic: time the slab of rank R of a W-GPU job on this one GPU")
    ap.add_argument("--extra-rows", action="store_true", help="also with one rank: the rows an N > 1 run adds (config 5's sweep per rank, the sharded tracking loop of config 4)")
    ap.add_argument("--no-loop", action="store_true", help="skip the full KinFu loop row (its single-frame integrate launches would enter a kernel-trace average of the headline kernel)")
    args = ap.parse_args()

    import torch

    from yak_amd import capi, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    emulate = None
    if args.as_rank:  # one GPU does the work of rank R of a W-rank job (same slab, same frame); no process group
        emulate = tuple(int(v) for v in args.as_rank.split(":"))
        args.gpus = world
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # rehearsal of the multi-rank path on a box with fewer GPUs than ranks (tests/test_bench_ranks_gpu.py):
    # YAK_BENCH_SHARE_GPU=1 puts every rank on GPU 0 and talks over gloo (RCCL refuses two ranks on one device)
    share_gpu = os.environ.get("YAK_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    # (YAK_BENCH_FORCE_PG=1: the process group also for one rank -- the RCCL branch rehearsed on a one-GPU box,
    # tests/test_bench_ranks_gpu.py)
    if world > 1 or os.environ.get("YAK_BENCH_FORCE_PG") == "1":
        import torch.distributed as dist

        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    # ---- workload ---------------------------------------------------------------------------------
    if emulate:
        world, rank = emulate
    if args.config == "c5":
        if world not in (1, 2, 4, 8):
            sys.exit("config c5: supported world sizes 1, 2, 4, 8")
        return run_c5(args, torch, capi, synth, dist, world, rank, local_rank, share_gpu, emulate)
    if world not in GLOBAL_DIMS:
        sys.exit("supported world sizes: %s" % sorted(GLOBAL_DIMS))
    dims = GLOBAL_DIMS[world]
    voxel = synth.VOLUME_SIZE_M / VOXELS_PER_RANK_EDGE
    _, _, tr = synth.volume_geometry(VOXELS_PER_RANK_EDGE)
    vs = (voxel, voxel, voxel)
    z0, z1 = slab_range(dims[2], world, rank)
    slab_voxels = dims[0] * dims[1] * (z1 - z0)

    if args.scene == "s2" and world == 1:
        depth, R, t = synth.scene_s2(40)
    elif args.scene == "s1" and world == 1:
        depth, R, t = synth.scene_s1()
    else:
        depth, R, t = wall_scene(dims, voxel)
    intr = synth.INTR
    rows, cols = depth.shape

    ctx = capi.Context(local_rank)  # launches on torch's current stream
    if args.z_chunk:
        ctx.set_option("z_chunk", args.z_chunk)
    if args.vx:
        ctx.set_option("vx", args.vx)
    if args.tile:
        ctx.set_option("tile", args.tile)
    if args.zu:
        ctx.set_option("zu", args.zu)
    if args.no_brick:
        ctx.set_option("brick_test", 0)
    if args.stacked:
        ctx.set_option("tile_side", 0)
    if args.order is not None:
        ctx.set_option("chunk_order", args.order)
    if args.chunk_min is not None:
        ctx.set_option("chunk_min", args.chunk_min)
    if args.chunk_ramp is not None:
        ctx.set_option("chunk_ramp", args.chunk_ramp)
    if args.no_nt:
        ctx.set_option("nontemporal", 0)
    d_depth = torch.from_numpy(depth.view(np.int16)).cuda()
    d_dists = torch.empty_like(d_depth)
    ctx.compute_dists(d_depth.data_ptr(), cols * 2, d_dists.data_ptr(), cols * 2, rows, cols, intr)
    vol = torch.empty(slab_voxels, dtype=torch.int32, device="cuda")
    desc = capi.volume_desc(vol.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT, z0, z1)
    Rf = R.reshape(-1)

    def step():
        ctx.integrate(desc, d_dists.data_ptr(), cols * 2, rows, cols, Rf, t, intr)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Every integrate_kernel launch of this process carries the library's HIP event pair (recorded on the launch
    # stream), from the very first one: the per-launch series in the JSON line covers the timed K launches
    # themselves, and `kernel_ms` is their mean (what the rocprofv3 kernel-trace average must agree with; the
    # per-frame prepass kernel is outside the pair).
    ctx.set_option("kernel_timing", 1)
    before = []  # everything the GPU does before the W warm-up steps, in order, with wall seconds

    def pre(name, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        before.append({"what": name, "s": round(time.perf_counter() - t0, 5)})
        return r

    # voxels updated per step on this rank: count them once (from a cleared volume every updated
    # voxel ends with weight 1); the count is the same every step because the frame is the same
    pre("clear_volume", lambda: ctx.clear_volume(desc))
    pre("1 integrate step + count of updated voxels", step)
    updated = int(((vol >> 16) == 1).sum().item())
    # the kernel's bit-exactness rests on v_rcp + Newton + Markstein being IEEE '/' on the GPU at hand: check it
    # here, on this box, before any number is reported (tests/test_division_gpu.py is the same check)
    selftest = pre("yakhip_selftest_division", ctx.selftest_division)
    if selftest[1] or selftest[3]:
        sys.exit("division self-test failed on this GPU: %s" % selftest)

    # beside the spec peak: what a kernel with no arithmetic at all reaches on this box with the same access
    # pattern (read 4 B, write 4 B per voxel, in place, 16 B per lane) -- torch's in-place integer add.
    # Calibrated BEFORE the timed region (stated in `before_timing`): this is %d sweeps = ~20 ms of full-rate
    # traffic, which also takes the board through its load-onset transient (tools/launch_series.py: from a cold
    # start launches 12-45, i.e. 2.4-9 ms after the load begins, run 8-12 %% slower, whatever ran before idle).
    def calibrate():
        ceil0, ceil1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        vol.add_(0)
        ceil0.record()
        for _ in range(CEILING_SWEEPS):
            vol.add_(0)
        ceil1.record()
        torch.cuda.synchronize()
        return 8.0 * slab_voxels / (ceil0.elapsed_time(ceil1) / CEILING_SWEEPS * 1e-3) / 1e9

    rmw_ceiling = pre("in-place RMW ceiling calibration, %d torch add_ sweeps of the slab" % CEILING_SWEEPS, calibrate)

    for _ in range(args.warmup):
        step()
    sync()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    sync()
    elapsed = time.perf_counter() - t0
    step_ms = ev0.elapsed_time(ev1) / args.steps  # whole step on the device: prepass + integrate kernel + gaps

    # steady state, outside the timed region and reported separately: STEADY_LAUNCHES more launches of the same
    # step; the mean of the second half is what a fusion loop that has been running for a while sees
    for _ in range(STEADY_LAUNCHES):
        step()
    series, starts = ctx.kernel_times(1 + args.warmup + args.steps + STEADY_LAUNCHES)
    ctx.set_option("kernel_timing", 0)
    first_timed = 1 + args.warmup
    timed = series[first_timed:first_timed + args.steps]
    kernel_ms = sum(timed) / max(len(timed), 1)
    tail = series[first_timed + args.steps + STEADY_LAUNCHES // 2:]
    steady_ms = sum(tail) / max(len(tail), 1)
    sync()

    totals = torch.tensor([elapsed, float(updated), kernel_ms, step_ms, steady_ms], dtype=torch.float64,
                          device="cpu" if share_gpu else "cuda")
    if dist is not None:
        mx = totals.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = totals.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed, kernel_ms, step_ms, steady_ms = float(mx[0]), float(mx[2]), float(mx[3]), float(mx[4])
        updated_total = int(sm[1])
    else:
        updated_total = updated

    if rank == 0 or emulate:
        gvox = updated_total * args.steps / elapsed / 1e9
        # algorithmic bytes of ONE launch on ONE GPU: 4 B read + 4 B written per
        # updated voxel + one pass over the half ray-distance image
        alg_bytes = 8 * updated + 2 * rows * cols
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        workload = "%dx%dx%d volume (%d^3 voxels per GPU, z-slabs), one %dx%d synthetic depth frame per step, " \
                   "scene %s, integrate only" % (dims[0], dims[1], dims[2], VOXELS_PER_RANK_EDGE, cols, rows,
                                                 "S0 wall (every voxel updates)" if args.scene == "s0" or world > 1
                                                 else {"s1": "S1 sphere+wall", "s2": "S2 orbit view 40"}[args.scene])
        out = {
            "metric": "voxel updates/sec",
            "value": round(gvox, 3),
            "unit": "Gvox/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload, "voxels_updated_per_step": updated_total,
                       "voxels_visited_per_step": dims[0] * dims[1] * dims[2], "voxel_bytes": 4, "parallelism": "z-slab x%d" % world},
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": load_traffic("integrate_%s_n%d" % (args.scene, world)),
                "kernel": "yakhip::integrate_kernel<4, true>",
                "kernel_ms": round(kernel_ms, 5),
                "algorithmic_bytes_per_launch": alg_bytes,
                "step_ms_device": round(step_ms, 5),
                "step_frac": round(alg_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                "measured_rmw_ceiling": {"GBps": round(rmw_ceiling, 1), "frac_of_it": round(achieved / rmw_ceiling, 4),
                                         "how": "torch in-place int32 add over the same slab (no arithmetic, same traffic), "
                                                "%d sweeps before the warm-up steps" % CEILING_SWEEPS},
                # the same kernel once the process has been running for a while: launches after the timed region
                "steady_state": {"kernel_ms": round(steady_ms, 5),
                                 "frac": round(alg_bytes / (steady_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                 "launches": "the last %d of %d launches issued after the timed region" % (
                                     len(tail), STEADY_LAUNCHES)},
                # device time of every integrate_kernel launch of this process (rank 0), in order: launch 0 is the
                # counting step on the cleared volume (general fuse path), then W warm-up, then the K timed launches
                "launch_series": {"first_timed_launch": first_timed, "kernel_ms": [round(v, 4) for v in series[:SERIES_LEN]],
                                  "start_ms": [round(v, 2) for v in starts[:SERIES_LEN]]},
                "before_timing": before,
            },
        }
        if emulate:
            out["config"]["emulated_rank"] = "%d of %d (this GPU's slab only: value is per GPU)" % (rank, world)
            out["n_gpus"] = 1
        if world == 1 and not args.no_cpu_baseline:
            dists_host = d_dists.cpu().numpy().view(np.uint16)
            out["cpu_baseline"] = cpu_baseline(dims, vs, tr, dists_host, R, t, intr, synth.MAX_WEIGHT)
        if world == 1 and not args.no_rows and args.scene == "s0":
            del vol
            try:
                out["next_rows"] = next_rows(ctx, torch, capi, synth, not args.no_cpu_baseline)
            except Exception as e:  # secondary: never take the headline down with it
                out["next_rows"] = {"error": "%s: %s" % (type(e).__name__, e)}
            if not args.no_loop:
                try:
                    out["orbit_views"] = orbit_views_row(ctx, torch, capi, synth)
                except Exception as e:
                    out["orbit_views"] = {"error": "%s: %s" % (type(e).__name__, e)}
                try:
                    out["kinfu_loop"] = kinfu_loop_row(ctx)
                except Exception as e:
                    out["kinfu_loop"] = {"error": "%s: %s" % (type(e).__name__, e)}
    # N > 1: the same command also carries north_star's scaling targets: BASELINE config 5 -- each rank's
    # 2048 x 2048 x 256 slab, four cameras per sweep, marching cubes -- and config 4, the sharded tracking loop.  All
    # ranks take part; rank 0 adds the rows to its line.  (YAK_BENCH_EXTRA=0 leaves them out.)
    extra = {}
    if ((dist is not None and world > 1 and not emulate and not args.no_rows and os.environ.get("YAK_BENCH_EXTRA", "1") != "0") or
            (args.extra_rows and not emulate)):
        vol = None
        torch.cuda.empty_cache()
        # These rows have never run on multi-GPU hardware (the pool gives one-GPU boxes): if a collective in them should
        # hang, the headline measured above must still get out.  A watchdog prints the line without them and ends the
        # process; it is cancelled when the rows are done.
        import threading

        def give_up():
            if rank == 0 or emulate:
                late = dict(out)
                late.update({k: v for k, v in extra.items() if v is not None})
                late["extra_rows"] = {"error": "timed out after %d s; rows finished by then are included" % extra_timeout}
                print(json.dumps(late), flush=True)
            os._exit(0)

        extra_timeout = int(os.environ.get("YAK_BENCH_EXTRA_TIMEOUT", "480"))
        watchdog = threading.Timer(extra_timeout, give_up)
        watchdog.daemon = True
        watchdog.start()
        for key, fn in (("c5", lambda: c5_measure(args, torch, capi, synth, dist, world, rank, local_rank, share_gpu, None,
                                                  steps=min(args.steps, 10), warmup=min(args.warmup, 3), with_cpu=False)),
                        ("kinfu_loop_sharded", lambda: kinfu_loop_sharded_row(ctx, torch, capi, synth, dist, world, rank))):
            try:
                extra[key] = fn()
            except Exception as e:  # secondary: never take the headline down with it (every rank fails alike or hangs in a
                extra[key] = {"error": "%s: %s" % (type(e).__name__, e)}  # collective: the watchdog above ends that)
        watchdog.cancel()
    if rank == 0 or emulate:
        for k, v in extra.items():
            if v is not None:
                out[k] = v
        print(json.dumps(out), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
