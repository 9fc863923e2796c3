"""The exchange steps of the sharded volume behind the C ABI (include/yakhip.h: yakhip_comm_*,
yakhip_raycast_sharded, yakhip_icp_estimate_sharded).  The test box has one GPU and RCCL accepts one rank per
GPU, so several ranks are played by THREADS of this process over the library's local transport (each rank its
own context and stream, its own slab + halo only), and the RCCL transport is exercised with a world of one --
the same ncclAllReduce calls on the context's stream.  A caller-supplied all-reduce (Python, over a barrier)
covers the third transport.

Bars: the sharded ray cast equals the whole-volume ray cast bit for bit on every rank; the sharded ICP estimate
gives every rank the same pose, within rounding of the one-GPU estimate (the order of the additions differs)."""
import threading

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from test_raycast_gpu import cam2vol, fused_volume, gpu_raycast  # noqa: E402
from yak_amd import capi, synth  # noqa: E402
from yak_amd.shard import held_range, slab_range  # noqa: E402

pytestmark = pytest.mark.gpu


def run_ranks(world, body):
    """body(rank, ctx, comm) on `world` threads over the local transport; re-raises the first failure."""
    comms = capi.Comm.local(world)
    errors, results = [], [None] * world

    def work(rank):
        ctx = None
        try:
            ctx = capi.Context(0, use_torch_stream=False)  # the context's own stream: ranks run concurrently
            results[rank] = body(rank, ctx, comms[rank])
            ctx.synchronize()
        except BaseException as e:  # noqa: BLE001
            errors.append((rank, e))
            comms[rank].close()  # breaks the group: the other ranks give up instead of waiting for ever
        finally:
            if ctx is not None:
                ctx.close()

    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=120)
    assert not any(th.is_alive() for th in threads), "a rank hangs"
    if errors:
        raise errors[0][1]
    for c in comms:
        c.close()
    return results


@pytest.mark.parametrize("world", [2, 3])
def test_local_transport_allreduce(world):
    rng = np.random.default_rng(5)
    f = rng.normal(size=(world, 1000)).astype(np.float32)
    k = rng.integers(-2 ** 31, 2 ** 31 - 1, size=(world, 777), dtype=np.int64).astype(np.int32)

    def body(rank, ctx, comm):
        assert comm.info() == (rank, world)
        a = torch.from_numpy(f[rank].copy()).cuda()
        b = torch.from_numpy(k[rank].copy()).cuda()
        c = torch.from_numpy(k[rank].copy()).cuda()
        torch.cuda.synchronize()
        ctx.allreduce(comm, a.data_ptr(), a.numel(), capi.SUM_F32)
        ctx.allreduce(comm, b.data_ptr(), b.numel(), capi.MIN_I32)
        ctx.allreduce(comm, c.data_ptr(), c.numel(), capi.SUM_I32)
        ctx.synchronize()
        return a.cpu().numpy(), b.cpu().numpy(), c.cpu().numpy()

    out = run_ranks(world, body)
    want = f[0].copy()
    for r in range(1, world):
        want = want + f[r]  # rank order, fp32
    for a, b, c in out:
        assert np.array_equal(a, want)  # the same bits on every rank
        assert np.array_equal(b, k.min(axis=0))
        assert np.array_equal(c, k.astype(np.int64).sum(axis=0).astype(np.int32))  # wraps like the device sum


def test_rccl_transport_world_of_one(gpu_ctx):
    """ncclGetUniqueId / ncclCommInitRank / ncclAllReduce through the library's lazily opened librccl."""
    uid = capi.Comm.unique_id()
    assert len(uid) == 128 and any(uid)
    comm = capi.Comm.rccl(gpu_ctx, uid, 1, 0)
    try:
        assert comm.info() == (0, 1)
        x = torch.arange(33, dtype=torch.float32, device="cuda") * 0.5
        gpu_ctx.allreduce(comm, x.data_ptr(), 33, capi.SUM_F32)
        k = torch.arange(-50, 50, dtype=torch.int32, device="cuda")
        gpu_ctx.allreduce(comm, k.data_ptr(), 100, capi.MIN_I32)
        torch.cuda.synchronize()
        assert torch.equal(x.cpu(), torch.arange(33, dtype=torch.float32) * 0.5)
        assert torch.equal(k.cpu(), torch.arange(-50, 50, dtype=torch.int32))
    finally:
        comm.close()


def _tracking_problem(n=96):
    vol, dims, vs, tr = fused_volume(n, [0, 40, 80], scene="s2")
    return vol, dims, vs, tr


def _sharded_raycast(rank, world, ctx, comm, vol, dims, vs, tr, R, t, Rinv, intr, rows, cols, want_depth=False, capacity=0):
    own = slab_range(dims[2], world, rank)
    halo = capi.raycast_halo(capi.volume_desc(256, dims, vs, tr, synth.MAX_WEIGHT))
    h0, h1 = held_range(dims[2], own[0], own[1], halo)
    plane = dims[0] * dims[1]
    held = torch.from_numpy(vol[h0 * plane:h1 * plane].view(np.int32).copy()).cuda()
    desc = capi.volume_desc(held.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT, h0, h1)
    keys = torch.full((rows, cols), -5, dtype=torch.int32, device="cuda")
    nrm = torch.full((rows, cols, 4), 7.0, dtype=torch.float32, device="cuda")
    pts = torch.full((rows, cols, 4), 7.0, dtype=torch.float32, device="cuda")
    dep = torch.full((rows, cols), 7, dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    if capacity:
        ctx.set_option("composite_capacity", capacity)
    kw = dict(depth_ptr=dep.data_ptr(), depth_pitch=cols * 2) if want_depth else dict(points_ptr=pts.data_ptr(),
                                                                                      points_pitch=cols * 16)
    ctx.raycast_sharded(comm, desc, own[0], own[1], R.reshape(-1), t, Rinv.reshape(-1), intr, keys.data_ptr(), cols * 4,
                        nrm.data_ptr(), cols * 16, rows, cols, **kw)
    status = ctx.comm_status(comm)
    return pts.cpu().numpy(), nrm.cpu().numpy(), dep.cpu().numpy(), keys.cpu().numpy(), status


@pytest.mark.parametrize("n,world,view", [(64, 2, 10), (96, 3, 77), (128, 4, 250)])
def test_sharded_raycast_equals_the_whole_volume_on_every_rank(gpu_ctx, n, world, view):
    vol, dims, vs, tr = fused_volume(n, [0, 60, 120, 180, 240], scene="s2")
    rows, cols, intr = 120, 160, (145.0, 145.0, 79.5, 59.5)
    _, R_v2c, t_v2c = synth.scene_s2(view)
    R, t = cam2vol(R_v2c, t_v2c)
    Rinv = np.ascontiguousarray(R.T)
    whole_p, whole_n = gpu_raycast(gpu_ctx, vol, dims, vs, tr, R, t, Rinv, intr, rows, cols)
    whole_d, whole_nd = gpu_raycast(gpu_ctx, vol, dims, vs, tr, R, t, Rinv, intr, rows, cols, want_depth=True)
    assert (~np.isnan(whole_p[..., 0])).sum() > 1500

    out = run_ranks(world, lambda r, ctx, comm: _sharded_raycast(r, world, ctx, comm, vol, dims, vs, tr, R, t, Rinv, intr, rows,
                                                                 cols))
    settled = 0
    for pts, nrm, _, keys, status in out:
        assert np.array_equal(pts.view(np.uint32), whole_p.view(np.uint32)), np.argwhere(pts.view(np.uint32) != whole_p.view(np.uint32))[:8]
        assert np.array_equal(nrm.view(np.uint32), whole_n.view(np.uint32)), np.argwhere(nrm.view(np.uint32) != whole_n.view(np.uint32))[:8]
        assert not ((keys != 0x7FFFFFFF) & (keys & 1 == 1)).any()  # nothing left unresolved
        assert status[1] == 0
        settled = status[0]
    # the depth overload
    out = run_ranks(world, lambda r, ctx, comm: _sharded_raycast(r, world, ctx, comm, vol, dims, vs, tr, R, t, Rinv, intr, rows,
                                                                 cols, want_depth=True))
    for _, nrm, dep, _, _ in out:
        assert np.array_equal(dep.view(np.uint16), whole_d) and np.array_equal(nrm.view(np.uint32), whole_nd.view(np.uint32))
    test_sharded_raycast_equals_the_whole_volume_on_every_rank.settled = getattr(
        test_sharded_raycast_equals_the_whole_volume_on_every_rank, "settled", 0) + settled


def test_second_composite_step_was_exercised_and_reports_overflow(gpu_ctx):
    assert getattr(test_sharded_raycast_equals_the_whole_volume_on_every_rank, "settled", 0) > 0
    # a list of one slot cannot hold a frame's deferred pixels: the call says so (and the caller repeats the frame)
    vol, dims, vs, tr = fused_volume(128, [0, 60, 120, 180, 240], scene="s2")
    rows, cols, intr = 120, 160, (145.0, 145.0, 79.5, 59.5)
    _, R_v2c, t_v2c = synth.scene_s2(250)
    R, t = cam2vol(R_v2c, t_v2c)
    Rinv = np.ascontiguousarray(R.T)
    out = run_ranks(4, lambda r, ctx, comm: _sharded_raycast(r, 4, ctx, comm, vol, dims, vs, tr, R, t, Rinv, intr, rows, cols,
                                                             capacity=1))
    assert all(s[0] > 1 and s[1] == 1 for *_, s in out), [s for *_, s in out]


def test_rccl_world_of_one_sharded_calls_equal_the_one_gpu_calls(gpu_ctx):
    vol, dims, vs, tr = _tracking_problem(64)
    rows, cols, intr = 120, 160, (145.0, 145.0, 79.5, 59.5)
    _, R_v2c, t_v2c = synth.scene_s2(40)
    R, t = cam2vol(R_v2c, t_v2c)
    Rinv = np.ascontiguousarray(R.T)
    whole_p, whole_n = gpu_raycast(gpu_ctx, vol, dims, vs, tr, R, t, Rinv, intr, rows, cols)
    comm = capi.Comm.rccl(gpu_ctx, capi.Comm.unique_id(), 1, 0)
    try:
        pts, nrm, _, _, status = _sharded_raycast(0, 1, gpu_ctx, comm, vol, dims, vs, tr, R, t, Rinv, intr, rows, cols)
        assert np.array_equal(pts.view(np.uint32), whole_p.view(np.uint32))
        assert np.array_equal(nrm.view(np.uint32), whole_n.view(np.uint32)) and status == (0, 0)
    finally:
        comm.close()


def _icp_levels(maps):
    """maps: per level (vcurr, ncurr, vprev, nprev) device tensors -> the tuples capi wants (10/5/4 linearisations)."""
    iters = (10, 5, 4)
    return [(m[0].data_ptr(), m[1].data_ptr(), m[2].data_ptr(), m[3].data_ptr(), m[0].shape[1] * 16, m[0].shape[0], m[0].shape[1],
             iters[l]) for l, m in enumerate(maps)]


def _icp_maps(ctx):
    """A real tracking problem: current maps of orbit frame 42 against the model ray cast from frame 40's pose."""
    import oracle

    vol, dims, vs, tr = _tracking_problem(96)
    depth, _, _ = synth.scene_s2(42)
    _, R_v2c, t_v2c = synth.scene_s2(40)
    R, t = cam2vol(R_v2c, t_v2c)
    vprev, nprev, _, _ = oracle.raycast(vol, dims, vs, tr, R, t, np.ascontiguousarray(R.T), synth.INTR, synth.ROWS, synth.COLS,
                                        nthreads=8)
    host = []
    d = oracle.bilateral_filter(depth)
    vp, npv = vprev, nprev
    for l in range(3):
        li = tuple(np.float32(v) / np.float32(1 << l) for v in synth.INTR)
        vc, nc = oracle.compute_points_normals(d, li)
        host.append((vc, nc, vp, npv))
        if l < 2:
            d = oracle.depth_pyramid(d)
            vp, npv = oracle.resize_points_normals(vp, npv)
    return host


def test_sharded_icp_estimate_same_pose_on_every_rank_and_close_to_one_gpu(gpu_ctx):
    host = _icp_maps(gpu_ctx)
    dev = [[torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in lvl] for lvl in host]
    R0, t0 = np.eye(3, dtype=np.float32).reshape(-1), np.zeros(3, dtype=np.float32)
    want_R, want_t, ok = gpu_ctx.icp_estimate(R0, t0, synth.INTR, 0.1, np.radians(30.0), _icp_levels(dev))
    assert ok and np.linalg.norm(want_t) > 1e-3  # the camera did move between the frames

    def body(rank, ctx, comm):
        mine = [[torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in lvl] for lvl in host]
        torch.cuda.synchronize()
        out = ctx.icp_estimate_sharded(comm, R0, t0, synth.INTR, 0.1, np.radians(30.0), _icp_levels(mine))
        del mine
        return out

    for world in (2, 3, 8):
        out = run_ranks(world, body)
        for R, t, ok in out:
            assert ok
            assert R == out[0][0] and t == out[0][1]  # lockstep: identical bits on every rank
            assert np.abs(np.float32(R) - np.float32(want_R)).max() < 2e-4 and np.abs(np.float32(t) - np.float32(want_t)).max() < 2e-4
    # RCCL transport, a world of one: the schedule with real ncclAllReduce calls between the launches
    comm = capi.Comm.rccl(gpu_ctx, capi.Comm.unique_id(), 1, 0)
    try:
        R, t, ok = gpu_ctx.icp_estimate_sharded(comm, R0, t0, synth.INTR, 0.1, np.radians(30.0), _icp_levels(dev))
        assert ok and np.abs(np.float32(R) - np.float32(want_R)).max() < 2e-4 and np.abs(np.float32(t) - np.float32(want_t)).max() < 2e-4
    finally:
        comm.close()


def test_sharded_icp_degenerate_input_takes_the_host_path_on_every_rank(gpu_ctx):
    """Featureless maps (all NaN): the device solve is not certified, every rank falls back to the host-side
    determinant test alike and reports the reference's `return false`."""
    rows, cols = 60, 80
    nan = np.full((rows, cols, 4), np.nan, dtype=np.float32)

    def body(rank, ctx, comm):
        m = [torch.from_numpy(nan.copy()).cuda() for _ in range(4)]
        torch.cuda.synchronize()
        lv = [(m[0].data_ptr(), m[1].data_ptr(), m[2].data_ptr(), m[3].data_ptr(), cols * 16, rows, cols, 3)]
        return ctx.icp_estimate_sharded(comm, np.eye(3).reshape(-1), np.zeros(3), (70.0, 70.0, 40.0, 30.0), 0.1, 0.5, lv)

    for R, t, ok in run_ranks(2, body):
        assert not ok and R == [1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0]


def test_custom_transport(gpu_ctx):
    """A caller-supplied all-reduce: two thread ranks meet at a Python barrier and add through host memory."""
    world = 2
    barrier = threading.Barrier(world)
    box = [None] * world

    ctxs = [None] * world

    def make(rank):
        def fn(buf, count, op, stream):
            # the values are ready in stream order on the rank's stream; the library's blocking copies run there
            host = np.empty(count, dtype=np.int32)
            ctxs[rank].download(host.ctypes.data, buf, count * 4)
            box[rank] = host
            barrier.wait()
            a, b = box[0], box[1]
            if op == capi.SUM_F32:
                out = (a.view(np.float32) + b.view(np.float32)).view(np.int32)
            elif op == capi.MIN_I32:
                out = np.minimum(a, b)
            else:
                out = a + b
            out = np.ascontiguousarray(out)
            barrier.wait()
            ctxs[rank].upload(buf, out.ctypes.data, count * 4)
            return 0
        return fn

    results = [None] * world
    errors = []

    def work(rank):
        try:
            ctx = ctxs[rank] = capi.Context(0, use_torch_stream=False)
            comm = capi.Comm.custom(world, rank, make(rank))
            x = torch.full((33,), float(rank + 1), dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            ctx.allreduce(comm, x.data_ptr(), 33, capi.SUM_F32)
            ctx.synchronize()
            results[rank] = x.cpu().numpy()
            comm.close()
            ctx.close()
        except BaseException as e:  # noqa: BLE001
            errors.append(e)
            barrier.abort()

    ths = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for th in ths:
        th.start()
    for th in ths:
        th.join(timeout=60)
    assert not errors, errors
    assert all(np.array_equal(r, np.full(33, 3.0, dtype=np.float32)) for r in results)


# ---- the failure path: yakhip_comm_abort ------------------------------------
def test_abort_wakes_a_rank_waiting_in_the_local_transport():
    """Rank 0 sits in an all-reduce that rank 1 never joins; rank 1 aborts its communicator instead: rank 0 comes
    back with YAKHIP_EABORTED, and every later exchange on either communicator fails at once."""
    comms = capi.Comm.local(2)
    ctx = capi.Context(0, use_torch_stream=False)
    buf = torch.ones(16, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    seen = {}

    def waiter():
        try:
            ctx.allreduce(comms[0], buf.data_ptr(), buf.numel(), capi.SUM_I32)
            seen["rc"] = 0
        except capi.YakHipError as e:
            seen["rc"], seen["text"] = e.code, str(e)

    th = threading.Thread(target=waiter)
    th.start()
    th.join(timeout=0.5)
    assert th.is_alive()  # it waits for rank 1
    comms[1].abort()
    th.join(timeout=30)
    assert not th.is_alive(), "the waiting rank was not woken"
    assert seen["rc"] == capi.EABORTED and "left the group" in seen["text"]
    comms[1].abort()  # idempotent
    for c in comms:
        with pytest.raises(capi.YakHipError) as e:
            ctx.allreduce(c, buf.data_ptr(), buf.numel(), capi.SUM_I32)
        assert e.value.code == capi.EABORTED
    for c in comms:
        c.close()
    ctx.close()


def test_completed_exchange_survives_a_peer_leaving():
    """A rank that leaves (destroy) right after the last barrier of an exchange must not turn the exchange its slower
    peer has completed into an error: many short exchanges, each followed by an immediate close on one side."""
    for _ in range(40):
        comms = capi.Comm.local(2)
        out = [None, None]

        def work(rank):
            ctx = capi.Context(0, use_torch_stream=False)
            b = torch.full((8,), rank + 1, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            try:
                ctx.allreduce(comms[rank], b.data_ptr(), b.numel(), capi.SUM_I32)
                ctx.synchronize()
                out[rank] = b.cpu().numpy()
            except capi.YakHipError as e:
                out[rank] = e
            if rank == 0:
                comms[0].close()  # at once
            ctx.close()

        ths = [threading.Thread(target=work, args=(r,)) for r in range(2)]
        for th in ths:
            th.start()
        for th in ths:
            th.join(timeout=60)
        comms[1].close()
        for o in out:
            assert isinstance(o, np.ndarray) and (o == 3).all(), o


def test_rccl_abort_world_of_one(gpu_ctx):
    """ncclCommAbort through the lazily opened librccl: afterwards collectives fail, destroy frees what is left."""
    comm = capi.Comm.rccl(gpu_ctx, capi.Comm.unique_id(), 1, 0)
    buf = torch.arange(32, dtype=torch.int32, device="cuda")
    gpu_ctx.allreduce(comm, buf.data_ptr(), buf.numel(), capi.SUM_I32)
    gpu_ctx.synchronize()
    assert buf.cpu().tolist() == list(range(32))
    comm.abort()
    with pytest.raises(capi.YakHipError) as e:
        gpu_ctx.allreduce(comm, buf.data_ptr(), buf.numel(), capi.SUM_I32)
    assert e.value.code == capi.EABORTED and "aborted" in str(e.value)
    comm.abort()
    comm.close()
    # the library is still usable: a fresh communicator works
    again = capi.Comm.rccl(gpu_ctx, capi.Comm.unique_id(), 1, 0)
    gpu_ctx.allreduce(again, buf.data_ptr(), buf.numel(), capi.SUM_I32)
    gpu_ctx.synchronize()
    again.close()
