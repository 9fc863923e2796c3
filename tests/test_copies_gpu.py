"""yakhip_upload / download[_2d] (the reference's DeviceMemory[2D]::upload / download, device_memory.cpp:129-138,
247-256): blocking copies between pageable host memory and device memory, staged through the context's
pinned buffers in 8 MiB pieces.  Byte-exact round trips over the piece boundaries."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from yak_amd import capi  # noqa: E402

pytestmark = pytest.mark.gpu

MiB = 1 << 20


@pytest.mark.parametrize("nbytes", [1, 4097, 8 * MiB - 1, 8 * MiB, 8 * MiB + 1, 16 * MiB, 41 * MiB + 13])
def test_linear_round_trip(gpu_ctx, nbytes):
    rng = np.random.default_rng(nbytes)
    src = rng.integers(0, 256, nbytes, dtype=np.uint8)
    dev = torch.zeros(nbytes + 64, dtype=torch.uint8, device="cuda")
    gpu_ctx.upload(dev.data_ptr() + 16, src.ctypes.data, nbytes)
    got = dev.cpu().numpy()
    assert np.array_equal(got[16:16 + nbytes], src) and not got[:16].any() and not got[16 + nbytes:].any()
    back = np.full(nbytes + 32, 0xAB, dtype=np.uint8)
    gpu_ctx.download(back.ctypes.data + 8, dev.data_ptr() + 16, nbytes)
    assert np.array_equal(back[8:8 + nbytes], src) and (back[:8] == 0xAB).all() and (back[8 + nbytes:] == 0xAB).all()


@pytest.mark.parametrize("rows,width,host_pitch,dev_pitch", [
    (480, 1280, 1280, 1280),          # a depth frame, contiguous on both sides
    (480, 1280, 1408, 1536),          # pitched on both sides
    (7, 3, 5, 256),                   # tiny
    (4100, 5000, 5000, 5120),         # many pieces of whole rows
    (3, 9 * MiB + 5, 9 * MiB + 64, 9 * MiB + 256),   # a row longer than a staging buffer
])
def test_pitched_round_trip(gpu_ctx, rows, width, host_pitch, dev_pitch):
    rng = np.random.default_rng(rows * 31 + width)
    host = rng.integers(0, 256, (rows, host_pitch), dtype=np.uint8)
    dev = torch.full((rows, dev_pitch), 7, dtype=torch.uint8, device="cuda")
    gpu_ctx.upload_2d(dev.data_ptr(), dev_pitch, host.ctypes.data, host_pitch, width, rows)
    got = dev.cpu().numpy()
    assert np.array_equal(got[:, :width], host[:, :width])
    assert (got[:, width:] == 7).all()                                        # padding untouched
    back = np.full((rows, host_pitch), 0xCD, dtype=np.uint8)
    gpu_ctx.download_2d(back.ctypes.data, host_pitch, dev.data_ptr(), dev_pitch, width, rows)
    assert np.array_equal(back[:, :width], host[:, :width]) and (back[:, width:] == 0xCD).all()


def test_copy_argument_checks(gpu_ctx):
    dev = torch.zeros(64, dtype=torch.uint8, device="cuda")
    host = np.zeros(64, dtype=np.uint8)
    gpu_ctx.upload(dev.data_ptr(), host.ctypes.data, 0)                       # nothing to do
    with pytest.raises(capi.YakHipError):
        gpu_ctx.upload(0, host.ctypes.data, 8)
    with pytest.raises(capi.YakHipError):
        gpu_ctx.download(host.ctypes.data, 0, 8)
    with pytest.raises(capi.YakHipError):
        gpu_ctx.upload_2d(dev.data_ptr(), 4, host.ctypes.data, 8, 8, 2)       # row wider than the device pitch


def test_device_to_device_copies(gpu_ctx):
    """yakhip_copy / yakhip_copy_2d (DeviceMemory[2D]::copyTo, device_memory.cpp:105-114, 233-242)."""
    src = torch.arange(100003, dtype=torch.int32, device="cuda")
    dst = torch.zeros_like(src)
    gpu_ctx.copy(dst.data_ptr(), src.data_ptr(), src.numel() * 4)
    assert torch.equal(dst, src)
    a = torch.arange(37 * 64, dtype=torch.int16, device="cuda").reshape(37, 64)
    b = torch.full((37, 80), -1, dtype=torch.int16, device="cuda")
    gpu_ctx.copy_2d(b.data_ptr(), 160, a.data_ptr(), 128, 53 * 2, 37)         # 53 of 64 columns into a wider image
    assert torch.equal(b[:, :53], a[:, :53]) and (b[:, 53:] == -1).all()
    from yak_amd.capi import YakHipError
    with pytest.raises(YakHipError):
        gpu_ctx.copy_2d(b.data_ptr(), 100, a.data_ptr(), 128, 53 * 2, 37)     # pitch smaller than a row
    gpu_ctx.copy(dst.data_ptr(), src.data_ptr(), 0)                             # nothing to do
