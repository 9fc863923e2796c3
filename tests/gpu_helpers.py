"""Shared plumbing for the -m gpu parity tests: numpy in, HIP kernels through the C ABI, numpy out.
torch only owns the device buffers."""
import numpy as np
import torch

from yak_amd import capi


def dev_u16_image(host, pitch_elems=None):
    """Upload a (rows, cols) uint16 image; optional row pitch (elements) > cols to test pitched rows."""
    rows, cols = host.shape
    pitch = cols if pitch_elems is None else pitch_elems
    buf = torch.zeros((rows, pitch), dtype=torch.int16, device="cuda")
    buf[:, :cols] = torch.from_numpy(host.view(np.int16)).cuda()
    return buf


def gpu_compute_dists(ctx, depth, intr, pitch_elems=None, out_pitch_elems=None):
    rows, cols = depth.shape
    d = dev_u16_image(depth, pitch_elems)
    op = cols if out_pitch_elems is None else out_pitch_elems
    out = torch.zeros((rows, op), dtype=torch.int16, device="cuda")
    ctx.compute_dists(d.data_ptr(), d.stride(0) * 2, out.data_ptr(), op * 2, rows, cols, intr)
    torch.cuda.synchronize()
    return out[:, :cols].cpu().numpy().view(np.uint16).copy()


def gpu_integrate(ctx, vol_host, dims, vs, trunc, max_weight, dists_host, R, t, intr, z0=0, z1=None, frames=1,
                  dists_pitch_elems=None, misalign_words=0):
    """Run `frames` integrate calls of the same frame on a copy of vol_host (uint32, layers [z0,z1))."""
    z1 = dims[2] if z1 is None else z1
    n = dims[0] * dims[1] * (z1 - z0)
    assert vol_host.size == n
    store = torch.zeros(n + 8, dtype=torch.int32, device="cuda")
    vol = store[misalign_words:misalign_words + n]
    vol.copy_(torch.from_numpy(vol_host.view(np.int32)).cuda())
    rows, cols = dists_host.shape
    dd = dev_u16_image(dists_host, dists_pitch_elems)
    desc = capi.volume_desc(vol.data_ptr(), dims, vs, trunc, max_weight, z0, z1)
    for _ in range(frames):
        ctx.integrate(desc, dd.data_ptr(), dd.stride(0) * 2, rows, cols, np.asarray(R).reshape(-1), t, intr)
    torch.cuda.synchronize()
    return vol.cpu().numpy().view(np.uint32).copy()


def half_ulp_distance(a_bits, b_bits):
    """Distance in half-precision ulps between two arrays of half bit patterns (finite values)."""
    def key(b):
        b = b.astype(np.int32)
        return np.where(b & 0x8000, -(b & 0x7FFF), b & 0x7FFF)
    return np.abs(key(a_bits) - key(b_bits))
