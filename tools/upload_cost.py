"""Host time of one depth frame's upload (yakhip_upload_2d, 480 x 640 u16) from pageable memory, a different source
buffer every time as in a frame loop."""
import sys, time
import numpy as np, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from yak_amd import capi
ctx = capi.Context(0)
rows, cols = 480, 640
frames = [np.random.default_rng(i).integers(0, 65536, (rows, cols), dtype=np.uint16) for i in range(200)]
dev = torch.zeros((rows, cols), dtype=torch.int16, device="cuda")
ctx.upload_2d(dev.data_ptr(), cols * 2, frames[0].ctypes.data, cols * 2, cols * 2, rows)
torch.cuda.synchronize()
t0 = time.perf_counter()
for f in frames:
    ctx.upload_2d(dev.data_ptr(), cols * 2, f.ctypes.data, cols * 2, cols * 2, rows)
t1 = time.perf_counter()
print("upload_2d of a depth frame: %.1f us per call (blocking)" % ((t1 - t0) / len(frames) * 1e6))
stage = np.empty((rows, cols), dtype=np.uint16)
t0 = time.perf_counter()
for f in frames:
    np.copyto(stage, f)
t1 = time.perf_counter()
print("the host copy alone (numpy copyto into one buffer): %.1f us" % ((t1 - t0) / len(frames) * 1e6))
pin = torch.empty((rows, cols), dtype=torch.int16).pin_memory()
torch.cuda.synchronize()
t0 = time.perf_counter()
for f in frames:
    dev.copy_(pin, non_blocking=True)
    torch.cuda.synchronize()
t1 = time.perf_counter()
print("a DMA from pinned memory + synchronise (torch): %.1f us" % ((t1 - t0) / len(frames) * 1e6))
