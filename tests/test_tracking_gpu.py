"""GPU parity for the tracking side (through the C ABI) against the CPU oracle: bilateral filter,
depth truncation, depth pyramid, point/normal maps, 2x2 resize, one ICP linearisation.

Bars: integer / index work bit-exact (pyramid, truncation, ICP outcome counters); fp32 maps bit-exact
(same expression tree, no contraction); bilateral u16 output equal except where the GPU's expf and
glibc's differ by an ulp right at a rounding boundary (|diff| <= 1 on < 0.1 % of pixels, stated
below); ICP sums within 1e-4 relative of the largest entry (the oracle accumulates in double, the GPU
in fp32 trees -- SURVEY.md 8e sets this tolerance).  Oracle status: PARITY UNPINNED.
"""
import math

import numpy as np
import pytest

torch = pytest.importorskip("torch")

import oracle  # noqa: E402
from yak_amd import capi, synth  # noqa: E402

pytestmark = pytest.mark.gpu

ICP_REL_TOL = 1e-4
BILATERAL_MISMATCH_RATE = 1e-3


def dev_u16(host, pad=0):
    rows, cols = host.shape
    buf = torch.full((rows, cols + pad), 9, dtype=torch.int16, device="cuda")
    buf[:, :cols] = torch.from_numpy(host.view(np.int16)).cuda()
    return buf


def dev_f4(host, pad=0):
    rows, cols = host.shape[:2]
    buf = torch.full((rows, cols + pad, 4), 9.0, dtype=torch.float32, device="cuda")
    buf[:, :cols] = torch.from_numpy(host).cuda()
    return buf


def same_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def same_maps(got, want):
    """float4 maps: NaN lanes in the same places, every other lane the same bits."""
    gn, wn = np.isnan(got), np.isnan(want)
    return np.array_equal(gn, wn) and same_bits(np.where(wn, 0, got), np.where(wn, 0, want))


@pytest.mark.parametrize("frame,pad", [(0, 0), (77, 0), (150, 24)])
def test_bilateral(gpu_ctx, frame, pad):
    depth, _, _ = synth.scene_s2(frame, noise_mm=2.0)
    rows, cols = depth.shape
    src = dev_u16(depth, pad)
    dst = torch.full_like(src, 9)
    gpu_ctx.bilateral_filter(src.data_ptr(), src.stride(0) * 2, dst.data_ptr(), dst.stride(0) * 2, rows, cols, 7, 4.5, 0.04)
    torch.cuda.synchronize()
    out = dst.cpu().numpy()
    assert (out[:, cols:] == 9).all()
    got = out[:, :cols].view(np.uint16) if pad == 0 else np.ascontiguousarray(out[:, :cols]).view(np.uint16)
    want = oracle.bilateral_filter(depth, 7, 4.5, 0.04)
    diff = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert diff.max() <= 1
    assert (diff != 0).mean() <= BILATERAL_MISMATCH_RATE, (diff != 0).sum()


def test_bilateral_small_and_odd_kernels(gpu_ctx):
    depth, _, _ = synth.scene_s1()
    patch = np.ascontiguousarray(depth[100:167, 200:291])   # 67 x 91
    for ksz, ss, sd in [(1, 1.0, 0.01), (3, 1.5, 0.02), (5, 2.0, 0.1), (9, 6.0, 0.04)]:
        src = dev_u16(patch)
        dst = torch.zeros_like(src)
        gpu_ctx.bilateral_filter(src.data_ptr(), src.stride(0) * 2, dst.data_ptr(), dst.stride(0) * 2, 67, 91, ksz, ss, sd)
        torch.cuda.synchronize()
        got = dst.cpu().numpy().view(np.uint16)
        want = oracle.bilateral_filter(patch, ksz, ss, sd)
        diff = np.abs(got.astype(np.int32) - want.astype(np.int32))
        assert diff.max() <= 1 and (diff != 0).mean() <= 5e-3, (ksz, (diff != 0).sum())


def test_truncate_and_pyramid(gpu_ctx):
    depth, _, _ = synth.scene_s2(33, noise_mm=1.5)
    rows, cols = depth.shape
    d = dev_u16(depth, 8)
    gpu_ctx.truncate_depth(d.data_ptr(), d.stride(0) * 2, rows, cols, 3.5)
    torch.cuda.synchronize()
    want = oracle.truncate_depth(depth, 3.5)
    assert (want == 0).sum() > (depth == 0).sum()
    host = d.cpu().numpy()
    assert (host[:, cols:] == 9).all()
    assert np.array_equal(np.ascontiguousarray(host[:, :cols]).view(np.uint16), want)

    level = depth
    src = dev_u16(level)
    for _ in range(3):                                           # 640x480 -> 320x240 -> 160x120 -> 80x60
        r, c = level.shape
        dst = torch.zeros((r // 2, c // 2), dtype=torch.int16, device="cuda")
        gpu_ctx.depth_pyramid(src.data_ptr(), src.stride(0) * 2, r, c, dst.data_ptr(), dst.stride(0) * 2, 0.04)
        torch.cuda.synchronize()
        want = oracle.depth_pyramid(level, 0.04)
        assert np.array_equal(dst.cpu().numpy().view(np.uint16), want)
        level, src = want, dst
    # odd source size: the last row / column is dropped (imgproc.cpp:17)
    odd = np.ascontiguousarray(depth[:101, :75])
    src = dev_u16(odd)
    dst = torch.zeros((50, 37), dtype=torch.int16, device="cuda")
    gpu_ctx.depth_pyramid(src.data_ptr(), src.stride(0) * 2, 101, 75, dst.data_ptr(), dst.stride(0) * 2, 0.04)
    torch.cuda.synchronize()
    assert np.array_equal(dst.cpu().numpy().view(np.uint16), oracle.depth_pyramid(odd, 0.04))


def gpu_points_normals(ctx, depth, intr, pad=0):
    rows, cols = depth.shape
    d = dev_u16(depth)
    pts = torch.full((rows, cols + pad, 4), 9.0, dtype=torch.float32, device="cuda")
    nrm = torch.full_like(pts, 9.0)
    ctx.compute_points_normals(intr, d.data_ptr(), d.stride(0) * 2, rows, cols, pts.data_ptr(), pts.stride(0) * 4,
                               nrm.data_ptr(), nrm.stride(0) * 4)
    torch.cuda.synchronize()
    return pts, nrm


def test_points_normals_and_resize(gpu_ctx):
    depth, _, _ = synth.scene_s2(120, noise_mm=1.0)
    depth = oracle.bilateral_filter(depth)
    rows, cols = depth.shape
    pts, nrm = gpu_points_normals(gpu_ctx, depth, synth.INTR, pad=4)
    want_p, want_n = oracle.compute_points_normals(depth, synth.INTR)
    ph, nh = pts.cpu().numpy(), nrm.cpu().numpy()
    assert (ph[:, cols:] == 9).all() and (nh[:, cols:] == 9).all()
    assert same_maps(ph[:, :cols], want_p) and same_maps(nh[:, :cols], want_n)
    assert (~np.isnan(want_p[..., 0])).sum() > 50000

    # 2 x 2 resize down the pyramid, as KinFu does with the ray-cast maps (kinfu.cpp:301-303)
    cur_p, cur_n = want_p, want_n
    dp, dn = dev_f4(want_p), dev_f4(want_n)
    for _ in range(3):
        r, c = cur_p.shape[:2]
        op = torch.full((r // 2, c // 2, 4), 9.0, dtype=torch.float32, device="cuda")
        on = torch.full_like(op, 9.0)
        gpu_ctx.resize_points_normals(dp.data_ptr(), dp.stride(0) * 4, dn.data_ptr(), dn.stride(0) * 4, r, c,
                                      op.data_ptr(), op.stride(0) * 4, on.data_ptr(), on.stride(0) * 4)
        torch.cuda.synchronize()
        wp, wn = oracle.resize_points_normals(cur_p, cur_n)
        assert same_maps(op.cpu().numpy(), wp) and same_maps(on.cpu().numpy(), wn)
        cur_p, cur_n, dp, dn = wp, wn, op, on

    # ... and two levels in one launch (yakhip_resize_points_normals2): the bits of two calls, also where the
    # half-size image has odd dimensions (its last row / column has no quarter-size pixel above it)
    for r, c in ((want_p.shape[0], want_p.shape[1]), (38, 54), (37, 51), (5, 4)):
        src_p, src_n = np.ascontiguousarray(want_p[:r, :c]), np.ascontiguousarray(want_n[:r, :c])
        h_p, h_n = oracle.resize_points_normals(src_p, src_n)
        q_p, q_n = oracle.resize_points_normals(h_p, h_n)
        dp, dn = dev_f4(src_p), dev_f4(src_n)
        o1p = torch.full((r // 2, c // 2 + 3, 4), 9.0, dtype=torch.float32, device="cuda")      # pitched rows
        o1n, o2p = torch.full_like(o1p, 9.0), torch.full((r // 4, c // 4, 4), 9.0, dtype=torch.float32, device="cuda")
        o2n = torch.full_like(o2p, 9.0)
        gpu_ctx.resize_points_normals2(dp.data_ptr(), dp.stride(0) * 4, dn.data_ptr(), dn.stride(0) * 4, r, c,
                                       o1p.data_ptr(), o1p.stride(0) * 4, o1n.data_ptr(), o1n.stride(0) * 4,
                                       o2p.data_ptr(), o2p.stride(0) * 4, o2n.data_ptr(), o2n.stride(0) * 4)
        torch.cuda.synchronize()
        g1p, g1n = o1p.cpu().numpy(), o1n.cpu().numpy()
        assert (g1p[:, c // 2:] == 9).all() and (g1n[:, c // 2:] == 9).all()
        assert same_maps(g1p[:, :c // 2], h_p) and same_maps(g1n[:, :c // 2], h_n)
        assert same_maps(o2p.cpu().numpy(), q_p) and same_maps(o2n.cpu().numpy(), q_n)
    with pytest.raises(capi.YakHipError):
        gpu_ctx.resize_points_normals2(dp.data_ptr(), 64, dn.data_ptr(), 64, 3, 4, o1p.data_ptr(), 64, o1n.data_ptr(), 64,
                                       o2p.data_ptr(), 64, o2n.data_ptr(), 64)


def level_maps(frame, level):
    depth, R, t = synth.scene_s2(frame, noise_mm=1.0)
    d = oracle.bilateral_filter(depth)
    for _ in range(level):
        d = oracle.depth_pyramid(d)
    intr = tuple(np.float32(v) / np.float32(1 << level) for v in synth.INTR)
    p, n = oracle.compute_points_normals(d, intr)
    return p, n, R.astype(np.float64), t.astype(np.float64)


def check_icp(ctx, R, t, level, vcurr, ncurr, vprev, nprev, dist=0.1, angle=math.radians(30.0), pad=0):
    rows, cols = vcurr.shape[:2]
    bufs = [dev_f4(a, pad) for a in (vcurr, ncurr, vprev, nprev)]
    args = []
    for b in bufs:
        args += [b.data_ptr(), b.stride(0) * 4]
    sums, counts = ctx.icp_accumulate(np.asarray(R, dtype=np.float32).reshape(-1), np.asarray(t, dtype=np.float32),
                                      synth.INTR, level, dist, angle, *args, rows, cols)
    want_s, want_c = oracle.icp_accumulate(R, t, synth.INTR, level, dist, angle, vcurr, ncurr, vprev, nprev)
    assert counts == [int(v) for v in want_c], (counts, want_c)           # outcome classification: exact
    assert sum(counts) == rows * cols
    scale = np.abs(want_s).max()
    assert np.abs(np.asarray(sums, dtype=np.float64) - want_s).max() <= ICP_REL_TOL * scale
    # determinism: the same call again gives the same bits
    sums2, counts2 = ctx.icp_accumulate(np.asarray(R, dtype=np.float32).reshape(-1), np.asarray(t, dtype=np.float32),
                                        synth.INTR, level, dist, angle, *args, rows, cols)
    assert counts2 == counts and np.array_equal(np.float32(sums2).view(np.uint32), np.float32(sums).view(np.uint32))
    return np.asarray(sums, dtype=np.float64), counts


@pytest.mark.parametrize("level", [0, 1, 2])
def test_icp_accumulate_levels(gpu_ctx, level):
    vprev, nprev, _, _ = level_maps(20, level)
    vcurr, ncurr, _, _ = level_maps(22, level)
    _, counts = check_icp(gpu_ctx, np.eye(3), np.zeros(3), level, vcurr, ncurr, vprev, nprev)
    assert counts[0] > 0.1 * vcurr.shape[0] * vcurr.shape[1]
    assert all(c > 0 for c in counts[:5])                                # every early-out is exercised


def test_icp_accumulate_poses_thresholds_and_pitch(gpu_ctx):
    vprev, nprev, Rp, tp = level_maps(100, 1)
    vcurr, ncurr, Rc, tc = level_maps(103, 1)
    R_true = Rp @ Rc.T
    t_true = tp - R_true @ tc
    check_icp(gpu_ctx, R_true, t_true, 1, vcurr, ncurr, vprev, nprev, pad=3)
    check_icp(gpu_ctx, R_true, t_true, 1, vcurr, ncurr, vprev, nprev, dist=0.01, angle=math.radians(5.0))
    _, counts = check_icp(gpu_ctx, np.eye(3), np.float32([0, 0, -10.0]), 1, vcurr, ncurr, vprev, nprev)
    assert counts[0] == 0 and counts[2] > 0                              # everything behind the camera: code 80
    empty = np.full_like(vcurr, np.nan)
    sums, counts = check_icp(gpu_ctx, np.eye(3), np.zeros(3), 1, empty, empty, vprev, nprev)
    assert counts[1] == vcurr.shape[0] * vcurr.shape[1] and not sums.any()


def test_icp_gauss_newton_on_gpu_sums_recovers_motion(gpu_ctx):
    """The full coarse-to-fine schedule of projective_icp.cpp:195-230 (iterations 10/5/4 on levels
    2/1/0) driven by the GPU's normal equations converges to the known orbit motion."""
    prev = [level_maps(40, l) for l in range(3)]
    curr = [level_maps(42, l) for l in range(3)]
    Rp, tp, Rc, tc = prev[0][2], prev[0][3], curr[0][2], curr[0][3]
    R_true = Rp @ Rc.T
    t_true = tp - R_true @ tc
    dev = [[dev_f4(a) for a in (curr[l][0], curr[l][1], prev[l][0], prev[l][1])] for l in range(3)]
    R, t = np.eye(3), np.zeros(3)
    for level, iters in ((2, 10), (1, 5), (0, 4)):
        rows, cols = curr[level][0].shape[:2]
        args = []
        for b in dev[level]:
            args += [b.data_ptr(), b.stride(0) * 4]
        for _ in range(iters):
            sums, counts = gpu_ctx.icp_accumulate(np.float32(R).reshape(-1), np.float32(t), synth.INTR, level, 0.1,
                                                  math.radians(30.0), *args, rows, cols)
            A, b = np.zeros((6, 6)), np.zeros(6)
            k = 0
            for i in range(6):
                for j in range(i, 7):
                    if j == 6:
                        b[i] = sums[k]
                    else:
                        A[i, j] = A[j, i] = sums[k]
                    k += 1
            assert abs(np.linalg.det(A)) > 1e-15
            r = np.linalg.solve(A, b)
            th = np.linalg.norm(r[:3])
            K = np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]]) / max(th, 1e-30)
            Rinc = np.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * K @ K
            R, t = Rinc @ R, Rinc @ t + r[3:]
    ang = math.degrees(math.acos(min(1.0, (np.trace(R_true.T @ R) - 1) / 2)))
    assert ang < 0.05 and np.linalg.norm(t - t_true) < 0.005, (ang, t, t_true)


def test_tracking_rejects_bad_arguments(gpu_ctx):
    from yak_amd import capi

    img = torch.zeros((8, 8), dtype=torch.int16, device="cuda")
    with pytest.raises(capi.YakHipError) as e:    # in place
        gpu_ctx.bilateral_filter(img.data_ptr(), 16, img.data_ptr(), 16, 8, 8, 7, 4.5, 0.04)
    assert e.value.code == -1
    out = torch.zeros_like(img)
    with pytest.raises(capi.YakHipError) as e:    # pitch smaller than a row
        gpu_ctx.bilateral_filter(img.data_ptr(), 8, out.data_ptr(), 16, 8, 8, 7, 4.5, 0.04)
    assert e.value.code == -1
    with pytest.raises(capi.YakHipError) as e:    # 1 x 8 source has no half-size level
        gpu_ctx.depth_pyramid(img.data_ptr(), 16, 1, 8, out.data_ptr(), 16, 0.04)
    assert e.value.code == -1
    f4 = torch.zeros((8, 8, 4), dtype=torch.float32, device="cuda")
    with pytest.raises(capi.YakHipError) as e:    # misaligned float4 map
        gpu_ctx.compute_points_normals(synth.INTR, img.data_ptr(), 16, 8, 8, f4.data_ptr() + 8, 128, f4.data_ptr(), 128)
    assert e.value.code == -2
    with pytest.raises(capi.YakHipError) as e:
        gpu_ctx.icp_accumulate(np.eye(3).reshape(-1), np.zeros(3), synth.INTR, 99, 0.1, 0.5, f4.data_ptr(), 128,
                               f4.data_ptr(), 128, f4.data_ptr(), 128, f4.data_ptr(), 128, 8, 8)
    assert e.value.code == -1


def _gn_loop_python(ctx, dev, shapes, schedule, R0, t0):
    """The reference's control flow (projective_icp.cpp:195-231) with numpy as the host solver."""
    R, t = np.array(R0, dtype=np.float64), np.array(t0, dtype=np.float64)
    for level, iters in schedule:
        rows, cols = shapes[level]
        args = []
        for b in dev[level]:
            args += [b.data_ptr(), b.stride(0) * 4]
        for _ in range(iters):
            sums, _ = ctx.icp_accumulate(np.float32(R).reshape(-1), np.float32(t), synth.INTR, level, 0.1,
                                         math.radians(30.0), *args, rows, cols)
            A, b = np.zeros((6, 6)), np.zeros(6)
            k = 0
            for i in range(6):
                for j in range(i, 7):
                    if j == 6:
                        b[i] = sums[k]
                    else:
                        A[i, j] = A[j, i] = sums[k]
                    k += 1
            if abs(np.linalg.det(A)) < 1e-15:
                return R, t, False
            r = np.linalg.solve(A, b)
            rv = np.float32(r[:3]).astype(np.float64)
            th = np.linalg.norm(rv)
            K = np.array([[0, -rv[2], rv[1]], [rv[2], 0, -rv[0]], [-rv[1], rv[0], 0]]) / max(th, 1e-300)
            Rinc = np.float32(np.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * K @ K).astype(np.float64)
            R = np.float32(Rinc @ R).astype(np.float64)
            t = np.float32(Rinc @ t + np.float32(r[3:])).astype(np.float64)
    return R, t, True


def test_icp_estimate_device_loop_equals_host_loop(gpu_ctx):
    """yakhip_icp_estimate (all 19 linearisations chained on the device, one wait) against the same
    schedule driven from the host one linearisation at a time.  The two solvers (device Cholesky, numpy
    LU) agree to double rounding; poses are compared at 1e-5."""
    prev = [level_maps(60, l) for l in range(3)]
    curr = [level_maps(62, l) for l in range(3)]
    dev = [[dev_f4(a) for a in (curr[l][0], curr[l][1], prev[l][0], prev[l][1])] for l in range(3)]
    shapes = [curr[l][0].shape[:2] for l in range(3)]
    levels = [(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[0].stride(0) * 4, shapes[l][0],
               shapes[l][1], it) for l, (d, it) in enumerate(zip(dev, (4, 5, 10)))]
    Rd, td, ok = gpu_ctx.icp_estimate(np.eye(3).reshape(-1), np.zeros(3), synth.INTR, 0.1, math.radians(30.0), levels)
    Rh, th, okh = _gn_loop_python(gpu_ctx, dev, shapes, ((2, 10), (1, 5), (0, 4)), np.eye(3), np.zeros(3))
    assert ok and okh
    assert np.abs(np.array(Rd).reshape(3, 3) - Rh).max() < 1e-5 and np.abs(np.array(td) - th).max() < 1e-5
    # and it is the true motion
    Rp, tp, Rc, tc = prev[0][2], prev[0][3], curr[0][2], curr[0][3]
    R_true = Rp @ Rc.T
    t_true = tp - R_true @ tc
    assert np.abs(np.array(Rd).reshape(3, 3) - R_true).max() < 2e-3 and np.linalg.norm(np.array(td) - t_true) < 5e-3
    # a non-identity start (pose hint) and a shorter schedule
    hint_R, hint_t = R_true, t_true + np.array([0.01, -0.005, 0.008])
    lv2 = [lv[:7] + (it,) for lv, it in zip(levels, (2, 0, 3))]
    Rd, td, ok = gpu_ctx.icp_estimate(np.float32(hint_R).reshape(-1), np.float32(hint_t), synth.INTR, 0.1,
                                      math.radians(30.0), lv2)
    Rh, th, okh = _gn_loop_python(gpu_ctx, dev, shapes, ((2, 3), (0, 2)), np.float32(hint_R), np.float32(hint_t))
    assert ok and okh
    assert np.abs(np.array(Rd).reshape(3, 3) - Rh).max() < 1e-5 and np.abs(np.array(td) - th).max() < 1e-5
    # deterministic
    Rd2, td2, _ = gpu_ctx.icp_estimate(np.float32(hint_R).reshape(-1), np.float32(hint_t), synth.INTR, 0.1,
                                       math.radians(30.0), lv2)
    assert Rd2 == Rd and td2 == td


def test_icp_estimate_weakly_constrained_view_stays_on_the_device(gpu_ctx):
    """Frames 250-260 of the orbit: the eigenvalues of the normal matrix are spread so far that the device
    solve's first certificate (AM-GM bound on lambda_min, icp.hip solve_update) fails although lambda_min is
    two orders of magnitude above OpenCV's cut-off; the second one (1 / trace(A^-1)) certifies the Cholesky
    solve.  Either way the pose must equal the host loop's."""
    prev = [level_maps(254, l) for l in range(3)]
    curr = [level_maps(256, l) for l in range(3)]
    dev = [[dev_f4(a) for a in (curr[l][0], curr[l][1], prev[l][0], prev[l][1])] for l in range(3)]
    shapes = [curr[l][0].shape[:2] for l in range(3)]
    # the case is what it claims to be (first linearisation of the coarsest level, GPU sums)
    args = []
    for b in dev[2]:
        args += [b.data_ptr(), b.stride(0) * 4]
    sums, _ = gpu_ctx.icp_accumulate(np.eye(3, dtype=np.float32).reshape(-1), np.zeros(3, dtype=np.float32), synth.INTR, 2,
                                     0.1, math.radians(30.0), *args, *shapes[2])
    A, k = np.zeros((6, 6)), 0
    for i in range(6):
        for j in range(i, 7):
            if j < 6:
                A[i, j] = A[j, i] = sums[k]
            k += 1
    w = np.linalg.eigvalsh(A)
    cut = 4.0 * 2.0 * 1.1920928955078125e-7 * np.trace(A)
    assert np.prod(w) / (np.trace(A) / 5.0) ** 5 < cut          # AM-GM bound: not certified
    assert 1.0 / np.sum(1.0 / w) > 10.0 * cut                   # tight bound: certified with room
    levels = [(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[0].stride(0) * 4, shapes[l][0],
               shapes[l][1], it) for l, (d, it) in enumerate(zip(dev, (4, 5, 10)))]
    Rd, td, ok = gpu_ctx.icp_estimate(np.eye(3).reshape(-1), np.zeros(3), synth.INTR, 0.1, math.radians(30.0), levels)
    Rh, th, okh = _gn_loop_python(gpu_ctx, dev, shapes, ((2, 10), (1, 5), (0, 4)), np.eye(3), np.zeros(3))
    assert ok and okh
    assert np.abs(np.array(Rd).reshape(3, 3) - Rh).max() < 1e-5 and np.abs(np.array(td) - th).max() < 1e-5
    Rp, tp, Rc, tc = prev[0][2], prev[0][3], curr[0][2], curr[0][3]
    R_true = Rp @ Rc.T
    t_true = tp - R_true @ tc
    assert np.abs(np.array(Rd).reshape(3, 3) - R_true).max() < 2e-3 and np.linalg.norm(np.array(td) - t_true) < 5e-3


def test_icp_estimate_degenerate_inputs_take_the_host_path(gpu_ctx):
    # no data at all: the normal matrix is zero, the reference's determinant test fails the frame
    p, n, _, _ = level_maps(10, 2)
    empty = np.full_like(p, np.nan)
    de, dp, dn = dev_f4(empty), dev_f4(p), dev_f4(n)
    rows, cols = p.shape[:2]
    lv = [(de.data_ptr(), de.data_ptr(), dp.data_ptr(), dn.data_ptr(), de.stride(0) * 4, rows, cols, 3)]
    R0 = np.float32([[0.999, 0, 0.04], [0, 1, 0], [-0.04, 0, 0.999]])
    Rd, td, ok = gpu_ctx.icp_estimate(R0.reshape(-1), np.float32([0.1, 0.2, 0.3]), synth.INTR, 0.1, 0.5, lv)
    assert not ok
    assert np.array_equal(np.float32(Rd).reshape(3, 3), R0) and td == [np.float32(0.1), np.float32(0.2), np.float32(0.3)]
    # a single fronto-parallel plane constrains 3 of 6 degrees of freedom: singular system; whatever the
    # determinant's rounding makes of it, the call must come back with a finite pose or ok == False
    depth = np.full((120, 160), 2000, dtype=np.uint16)
    pp, pn = oracle.compute_points_normals(depth, tuple(np.float32(v) / 4 for v in synth.INTR))
    d1, d2 = dev_f4(pp), dev_f4(pn)
    lv = [(d1.data_ptr(), d2.data_ptr(), d1.data_ptr(), d2.data_ptr(), d1.stride(0) * 4, 120, 160, 2)]
    # level index 0 with quarter-size maps: pass the matching intrinsics as "level 0"
    Rd, td, ok = gpu_ctx.icp_estimate(np.eye(3).reshape(-1), np.zeros(3), tuple(v / 4 for v in synth.INTR), 0.1, 0.5, lv)
    assert (not ok) or (np.isfinite(Rd).all() and np.isfinite(td).all())


def test_icp_estimate_schedule_edges(gpu_ctx):
    """The launch chain at its ends: no linearisation at all (the start pose comes back untouched), exactly
    one (launch 0 + the closing solve), two (the first launch that finishes its predecessor)."""
    prev, curr = level_maps(60, 1), level_maps(62, 1)
    dev = {1: [dev_f4(a) for a in (curr[0], curr[1], prev[0], prev[1])]}
    rows, cols = curr[0].shape[:2]
    d = dev[1]
    R0 = np.float32([[0.9998, 0, 0.02], [0, 1, 0], [-0.02, 0, 0.9998]])
    t0 = np.float32([0.01, -0.02, 0.03])

    def lv(n):  # pyramid level 1 holds the maps, level 0 is present but idle
        return [(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), 16, 1, 1, 0),
                (d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[0].stride(0) * 4, rows, cols, n)]

    Rd, td, ok = gpu_ctx.icp_estimate(R0.reshape(-1), t0, synth.INTR, 0.1, math.radians(30.0), lv(0))
    assert ok and np.array_equal(np.float32(Rd).reshape(3, 3), R0) and np.array_equal(np.float32(td), t0)
    for n in (1, 2, 3):
        Rd, td, ok = gpu_ctx.icp_estimate(np.eye(3).reshape(-1), np.zeros(3), synth.INTR, 0.1, math.radians(30.0), lv(n))
        Rh, th, okh = _gn_loop_python(gpu_ctx, dev, {1: (rows, cols)}, ((1, n),), np.eye(3), np.zeros(3))
        assert ok and okh
        assert np.abs(np.array(Rd).reshape(3, 3) - Rh).max() < 1e-5 and np.abs(np.array(td) - th).max() < 1e-5, n


def test_fusion_loop_with_the_side_stream_gives_the_same_volume_and_poses(gpu_ctx):
    """The loop of yak_amd/loop.py prepares frame f + 1 on the context's second stream while frame f is still being
    integrated and ray cast (yakhip_ctx_side_begin / _end, two ray-distance images in turn): the same voxels and
    the same poses as everything in one stream, tracking from depth alone and with supplied poses."""
    from yak_amd.loop import run_kinfu_loop

    scene = [synth.scene_s2(f, noise_mm=1.0) for f in range(0, 24)]
    for icp in ("device", None):
        a = run_kinfu_loop(gpu_ctx, 128, 24, icp, scene, overlap=True)
        b = run_kinfu_loop(gpu_ctx, 128, 24, icp, scene, overlap=False)
        c = run_kinfu_loop(gpu_ctx, 128, 24, icp, scene, overlap=True, surface_map=False)
        assert a["volume_checksum"] == b["volume_checksum"] == c["volume_checksum"] != 0
        assert a["final_pose"] == b["final_pose"] == c["final_pose"]
    with pytest.raises(capi.YakHipError):
        gpu_ctx.side_end()                                         # not on the side stream
    gpu_ctx.side_begin()
    gpu_ctx.side_begin()                                           # a second begin goes on with the open section
    gpu_ctx.side_end()
    with pytest.raises(capi.YakHipError):
        gpu_ctx.side_end()


@pytest.mark.parametrize("rows,cols", [(2, 2), (3, 5), (2, 67), (33, 2), (480, 640)])
@pytest.mark.parametrize("content", ["zeros", "max", "full range"])
def test_preprocessing_extreme_depths_and_tiny_images(gpu_ctx, rows, cols, content):
    """No return anywhere, the largest depth everywhere, every 16-bit value: truncation and pyramid exact, the
    bilateral filter within its stated bar, point / normal maps the same bits, down to a 2 x 2 image."""
    rng = np.random.default_rng(rows * 1000 + cols)
    depth = {"zeros": np.zeros((rows, cols), dtype=np.uint16), "max": np.full((rows, cols), 65535, dtype=np.uint16),
             "full range": rng.integers(0, 65536, size=(rows, cols), dtype=np.uint16)}[content]
    src = dev_u16(depth, 3)
    dst = torch.full_like(src, 9)
    gpu_ctx.bilateral_filter(src.data_ptr(), src.stride(0) * 2, dst.data_ptr(), dst.stride(0) * 2, rows, cols, 7, 4.5, 0.04)
    torch.cuda.synchronize()
    out = dst.cpu().numpy()
    assert (out[:, cols:] == 9).all()
    got = np.ascontiguousarray(out[:, :cols]).view(np.uint16)
    want = oracle.bilateral_filter(depth, 7, 4.5, 0.04)
    diff = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert diff.max() <= 1 and (diff != 0).mean() <= max(BILATERAL_MISMATCH_RATE, 1.0 / (rows * cols) if rows * cols < 1000 else 0)

    d = dev_u16(depth)
    gpu_ctx.truncate_depth(d.data_ptr(), d.stride(0) * 2, rows, cols, 30.0)
    torch.cuda.synchronize()
    assert np.array_equal(d.cpu().numpy().view(np.uint16), oracle.truncate_depth(depth, 30.0))

    half = torch.full((rows // 2, cols // 2), 9, dtype=torch.int16, device="cuda")
    s = dev_u16(depth)
    gpu_ctx.depth_pyramid(s.data_ptr(), s.stride(0) * 2, rows, cols, half.data_ptr(), half.stride(0) * 2, 0.04)
    torch.cuda.synchronize()
    assert np.array_equal(half.cpu().numpy().view(np.uint16), oracle.depth_pyramid(depth, 0.04))

    intr = (0.9 * max(cols, 4), 0.9 * max(cols, 4), cols / 2 - 0.5, rows / 2 - 0.5)
    pts, nrm = gpu_points_normals(gpu_ctx, depth, intr, pad=2)
    want_p, want_n = oracle.compute_points_normals(depth, intr)
    ph, nh = pts.cpu().numpy(), nrm.cpu().numpy()
    assert (ph[:, cols:] == 9).all() and (nh[:, cols:] == 9).all()
    assert same_maps(ph[:, :cols], want_p) and same_maps(nh[:, :cols], want_n)


ICP_POSES = {
    "NaN translation": (np.eye(3), [0.0, np.nan, 0.0]),
    "NaN rotation entry": (np.where(np.eye(3) > 0, 1.0, np.nan), [0.0, 0.0, 0.0]),
    "infinite translation": (np.eye(3), [0.0, 0.0, np.inf]),
    "scaled rotation": (1.3 * np.eye(3), [0.01, 0.0, 0.02]),
    "mirrored": (np.diag([-1.0, 1.0, 1.0]), [0.0, 0.0, 0.0]),
    "zero rotation": (np.zeros((3, 3)), [0.0, 0.0, 1.5]),
    "far away": (np.eye(3), [0.0, 0.0, 1e6]),
}


@pytest.mark.parametrize("case", sorted(ICP_POSES))
def test_icp_accumulate_degenerate_poses(gpu_ctx, case):
    """Poses no track produces: every pixel must land in the oracle's outcome class (find_coresp, proj_icp.cu:74-102,
    with IEEE inf / NaN arithmetic) and the sums must agree to the usual bar."""
    vprev, nprev, _, _ = level_maps(20, 1)
    vcurr, ncurr, _, _ = level_maps(22, 1)
    R, t = ICP_POSES[case]
    check_icp(gpu_ctx, np.asarray(R, dtype=np.float64), np.asarray(t, dtype=np.float64), 1, vcurr, ncurr, vprev, nprev)
