"""GPU parity for the debug display (yakhip_render_image_points / _depth / yakhip_render_tangent_colors,
through the C ABI) against the CPU oracle: every pixel the same four bytes (integer output of an fp32
expression tree restated statement by statement, no contraction, correctly rounded sqrt and divide).
Oracle status: PARITY UNPINNED (no image in the reference; __powf restated as v^20 by multiplication)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

import oracle  # noqa: E402
from yak_amd import synth  # noqa: E402

pytestmark = pytest.mark.gpu

LIGHTS = [(0.0, 0.0, 0.0), (0.5, -0.3, -1.0), (-2.0, 1.0, 0.5)]


def frame_maps(frame, step=1):
    depth, _, _ = synth.scene_s2(frame, noise_mm=1.0)
    depth = np.ascontiguousarray(depth[::step, ::step])
    intr = [v / step for v in synth.INTR]
    pts, nrm = oracle.compute_points_normals(depth, intr)
    return depth, pts, nrm, intr


def dev_f4(host, pad=0):
    rows, cols = host.shape[:2]
    buf = torch.full((rows, cols + pad, 4), 9.0, dtype=torch.float32, device="cuda")
    buf[:, :cols] = torch.from_numpy(host).cuda()
    return buf


@pytest.mark.parametrize("frame,pad,light", [(0, 0, LIGHTS[0]), (60, 8, LIGHTS[1]), (140, 3, LIGHTS[2])])
def test_render_points(gpu_ctx, frame, pad, light):
    _, pts, nrm, _ = frame_maps(frame)
    rows, cols = pts.shape[:2]
    dp, dn = dev_f4(pts, pad), dev_f4(nrm)
    img = torch.full((rows, cols + pad), 0x55555555, dtype=torch.int32, device="cuda")
    gpu_ctx.render_image_points(dp.data_ptr(), dp.stride(0) * 4, dn.data_ptr(), dn.stride(0) * 4, rows, cols, light,
                                img.data_ptr(), img.stride(0) * 4)
    torch.cuda.synchronize()
    host = img.cpu().numpy()
    assert (host[:, cols:] == 0x55555555).all()
    got = np.ascontiguousarray(host[:, :cols]).view(np.uint8).reshape(rows, cols, 4)
    want = oracle.render_image(pts, nrm, light)
    assert np.array_equal(got, want), int((got != want).any(axis=2).sum())
    assert (~np.isnan(pts[..., 0])).sum() > 50000 and np.isnan(pts[..., 0]).sum() > 1000


def test_render_depth_with_the_reference_wrappers_intrinsics(gpu_ctx):
    depth, _, nrm, intr = frame_maps(25)
    rows, cols = depth.shape
    quirk = [intr[0], intr[1], intr[2], intr[1]]                  # imgproc.cpp:92: (fx, fy, cx, fy)
    for args in (intr, quirk):
        dd = torch.from_numpy(depth.view(np.int16)).cuda()
        dn = dev_f4(nrm)
        img = torch.zeros((rows, cols), dtype=torch.int32, device="cuda")
        gpu_ctx.render_image_depth(dd.data_ptr(), cols * 2, dn.data_ptr(), dn.stride(0) * 4, rows, cols, args, LIGHTS[1],
                                   img.data_ptr(), cols * 4)
        torch.cuda.synchronize()
        got = img.cpu().numpy().view(np.uint8).reshape(rows, cols, 4)
        assert np.array_equal(got, oracle.render_image(depth, nrm, LIGHTS[1], intr=args))


def test_tangent_colours_and_side_by_side_layout(gpu_ctx):
    """KinFu::renderImage(flag 3) renders two views into the halves of one image (kinfu.cpp:328-332): the
    second view's base pointer is offset by `cols` pixels and both share the wide pitch."""
    _, pts, nrm, _ = frame_maps(100, step=2)                       # 240 x 320
    nrm[5, 7, :3] = (3.0, -4.0, 2.5)                               # not a unit normal: bytes saturate
    nrm[6, 7, :3] = (-3.0, 4.0, -2.5)
    rows, cols = pts.shape[:2]
    dp, dn = dev_f4(pts), dev_f4(nrm)
    wide = torch.zeros((rows, 2 * cols), dtype=torch.int32, device="cuda")
    gpu_ctx.render_image_points(dp.data_ptr(), dp.stride(0) * 4, dn.data_ptr(), dn.stride(0) * 4, rows, cols, LIGHTS[1],
                                wide.data_ptr(), wide.stride(0) * 4)
    gpu_ctx.render_tangent_colors(dn.data_ptr(), dn.stride(0) * 4, rows, cols, wide.data_ptr() + cols * 4,
                                  wide.stride(0) * 4)
    torch.cuda.synchronize()
    got = wide.cpu().numpy().view(np.uint8).reshape(rows, 2 * cols, 4)
    assert np.array_equal(got[:, :cols], oracle.render_image(pts, nrm, LIGHTS[1]))
    want = oracle.render_tangent_colors(nrm)
    assert np.array_equal(got[:, cols:], want)
    assert want[5, 7].tolist() == [0, 255, 0, 0] and want[6, 7].tolist() == [255, 0, 255, 0]


def test_render_degenerate_pixels(gpu_ctx):
    """NaN normals under valid points, a point on the light, a point at the eye, zero-length normals."""
    pts = np.zeros((4, 64, 4), dtype=np.float32)
    nrm = np.zeros_like(pts)
    pts[..., 2] = 1.0
    pts[0, :, 0] = np.linspace(-1, 1, 64)
    nrm[..., 2] = -1.0
    nrm[1, :, :3] = np.nan                      # both Phong terms drop out (fmax(0, NaN) = 0): ambient only
    pts[2, 5, :3] = (0.5, -0.3, -1.0)           # on the light
    pts[2, 6, :3] = 0.0                         # at the eye
    nrm[3, :, :3] = 0.0                         # zero normal: R = -L
    pts[3, 7, 0] = np.nan                       # backdrop
    dp, dn = dev_f4(pts), dev_f4(nrm)
    img = torch.zeros((4, 64), dtype=torch.int32, device="cuda")
    gpu_ctx.render_image_points(dp.data_ptr(), 64 * 16, dn.data_ptr(), 64 * 16, 4, 64, LIGHTS[1], img.data_ptr(), 64 * 4)
    tan = torch.zeros((4, 64), dtype=torch.int32, device="cuda")
    gpu_ctx.render_tangent_colors(dn.data_ptr(), 64 * 16, 4, 64, tan.data_ptr(), 64 * 4)
    torch.cuda.synchronize()
    got = img.cpu().numpy().view(np.uint8).reshape(4, 64, 4)
    want = oracle.render_image(pts, nrm, LIGHTS[1])
    assert np.array_equal(got, want)
    assert (want[1, :, :3] == 76).all() and want[2, 5].tolist() == [76, 76, 76, 0]
    assert np.array_equal(tan.cpu().numpy().view(np.uint8).reshape(4, 64, 4), oracle.render_tangent_colors(nrm))


def test_render_rejects_bad_arguments(gpu_ctx):
    from yak_amd.capi import YakHipError
    buf = torch.zeros((8, 8, 4), dtype=torch.float32, device="cuda")
    img = torch.zeros((8, 8), dtype=torch.int32, device="cuda")
    with pytest.raises(YakHipError):                               # pitch smaller than a row
        gpu_ctx.render_image_points(buf.data_ptr(), 8 * 16, buf.data_ptr(), 8 * 16, 8, 8, LIGHTS[0], img.data_ptr(), 16)
    with pytest.raises(YakHipError):                               # zero focal length
        gpu_ctx.render_image_depth(img.data_ptr(), 16, buf.data_ptr(), 8 * 16, 8, 8, [0, 1, 0, 0], LIGHTS[0],
                                   img.data_ptr(), 32)
    with pytest.raises(YakHipError):                               # empty image
        gpu_ctx.render_tangent_colors(buf.data_ptr(), 8 * 16, 0, 8, img.data_ptr(), 32)
