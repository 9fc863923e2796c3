"""Host time of a yakhip_integrate call (enqueue only), with and without a prepared frame: what stands between the ICP
result and the first wave of the integrate kernel in the frame loop."""
import time, numpy as np, torch, sys
sys.path.insert(0, "/root/repo")
from yak_amd import capi, synth
ctx = capi.Context(0)
dims, vs, tr = synth.volume_geometry(512)
depth, R, t = synth.scene_s2(40)
rows, cols = depth.shape
d = torch.from_numpy(depth.view(np.int16)).cuda()
dd = torch.empty_like(d)
ctx.compute_dists(d.data_ptr(), cols * 2, dd.data_ptr(), cols * 2, rows, cols, synth.INTR)
vol = torch.zeros(512 ** 3, dtype=torch.int32, device="cuda")
desc = capi.volume_desc(vol.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT)
Rf = np.float32(R).reshape(-1); tf = np.float32(t)
lib = capi.load()
import ctypes as C
Rc, tc, ic = capi._f32(Rf, 9), capi._f32(tf, 3), capi._f32(synth.INTR, 4)
for _ in range(5):
    ctx.integrate(desc, dd.data_ptr(), cols * 2, rows, cols, Rf, tf, synth.INTR)
torch.cuda.synchronize()
for n in (50, 50):
    t0 = time.perf_counter()
    for _ in range(n):
        lib.yakhip_integrate(ctx._h, C.byref(desc), C.c_void_p(dd.data_ptr()), cols * 2, rows, cols, Rc, tc, ic)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("yakhip_integrate: %.2f us of host time per call (enqueue only), %.1f us per call to completion" % ((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
# with a prepared frame
for n in (50,):
    t0 = time.perf_counter()
    for _ in range(n):
        lib.yakhip_integrate_prepare(ctx._h, C.byref(desc), C.c_void_p(dd.data_ptr()), cols * 2, rows, cols)
        lib.yakhip_integrate(ctx._h, C.byref(desc), C.c_void_p(dd.data_ptr()), cols * 2, rows, cols, Rc, tc, ic)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("prepare + integrate: %.2f us of host time per pair" % ((t1 - t0) / n * 1e6))
tp = ti = 0.0
n = 50
for _ in range(n):
    a = time.perf_counter()
    lib.yakhip_integrate_prepare(ctx._h, C.byref(desc), C.c_void_p(dd.data_ptr()), cols * 2, rows, cols)
    b = time.perf_counter()
    lib.yakhip_integrate(ctx._h, C.byref(desc), C.c_void_p(dd.data_ptr()), cols * 2, rows, cols, Rc, tc, ic)
    c = time.perf_counter()
    tp += b - a
    ti += c - b
torch.cuda.synchronize()
print("prepare alone %.2f us, integrate of a prepared frame %.2f us (host time per call)" % (tp / n * 1e6, ti / n * 1e6))
# an empty ctypes call for scale
a = time.perf_counter()
for _ in range(1000):
    lib.yakhip_last_error()
print("a ctypes call that does nothing: %.2f us" % ((time.perf_counter() - a) / 1000 * 1e6))
