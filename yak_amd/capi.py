"""ctypes binding of libyakhip.so (include/yakhip.h) -- the thin Python side of the C ABI.

Used by tests/, bench.py and __graft_entry__.py to drive the HIP kernels with device memory
owned by torch (plumbing only: tensors supply `data_ptr()` and the current stream).
There is no CPU fallback: if the library is missing or no GPU is visible, calls raise.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libyakhip.so")

EXPORTS = [
    "yakhip_ctx_create", "yakhip_ctx_destroy", "yakhip_ctx_set_stream", "yakhip_ctx_use_own_stream", "yakhip_ctx_synchronize",
    "yakhip_device_count", "yakhip_device_name", "yakhip_last_error", "yakhip_ctx_set_option", "yakhip_ctx_side_begin", "yakhip_ctx_side_end", "yakhip_ctx_stream",
    "yakhip_ctx_kernel_time", "yakhip_ctx_kernel_times", "yakhip_ctx_brick_stats", "yakhip_ctx_launch_counts", "yakhip_selftest_division",
    "yakhip_malloc", "yakhip_malloc_pitch", "yakhip_free", "yakhip_upload", "yakhip_download",
    "yakhip_upload_2d", "yakhip_download_2d", "yakhip_copy", "yakhip_copy_2d",
    "yakhip_clear_volume", "yakhip_surface_map_bytes", "yakhip_surface_map_rebuild", "yakhip_compute_dists", "yakhip_integrate", "yakhip_integrate_static",
    "yakhip_integrate_batch",
    "yakhip_raycast_points", "yakhip_raycast_points_pyramid", "yakhip_raycast_depth", "yakhip_raycast_slab", "yakhip_raycast_resolve", "yakhip_raycast_halo",
    "yakhip_bilateral_filter", "yakhip_truncate_depth", "yakhip_depth_pyramid", "yakhip_compute_points_normals",
    "yakhip_integrate_prepare", "yakhip_resize_points_normals", "yakhip_resize_points_normals2", "yakhip_icp_accumulate", "yakhip_icp_accumulate_rows", "yakhip_icp_estimate",
    "yakhip_marching_cubes", "yakhip_marching_cubes_layers", "yakhip_weld_vertices", "yakhip_extract_cloud", "yakhip_extract_normals",
    "yakhip_render_image_points", "yakhip_render_image_depth", "yakhip_render_tangent_colors",
    "yakhip_comm_unique_id", "yakhip_comm_create_rccl", "yakhip_comm_create_local", "yakhip_comm_create_custom",
    "yakhip_comm_destroy", "yakhip_comm_abort", "yakhip_comm_info", "yakhip_comm_allreduce", "yakhip_comm_status",
    "yakhip_raycast_sharded", "yakhip_icp_estimate_sharded",
]

SUM_F32, MIN_I32, SUM_I32 = 0, 1, 2
EINVAL, EALIGN, ENOGPU, ERANGE, EABORTED = -1, -2, -3, -4, -5  # include/yakhip.h
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)


class YakHipError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("yakhip error %d: %s" % (code, text))
        self.code = code


class VolumeDesc(C.Structure):
    """struct yakhip_volume"""
    _fields_ = [
        ("data", C.c_void_p),
        ("dims", C.c_int * 3),
        ("z0", C.c_int),
        ("z1", C.c_int),
        ("voxel_size", C.c_float * 3),
        ("trunc_dist", C.c_float),
        ("max_weight", C.c_int),
        ("surface_map", C.c_void_p),
    ]


class FrameDesc(C.Structure):
    """struct yakhip_frame"""
    _fields_ = [
        ("dists", C.c_void_p),
        ("dists_pitch", C.c_size_t),
        ("rows", C.c_int),
        ("cols", C.c_int),
        ("R", C.c_float * 9),
        ("t", C.c_float * 3),
        ("intr", C.c_float * 4),
    ]


class IcpLevel(C.Structure):
    """struct yakhip_icp_level"""
    _fields_ = [
        ("vcurr", C.c_void_p), ("ncurr", C.c_void_p), ("vprev", C.c_void_p), ("nprev", C.c_void_p),
        ("vcurr_pitch", C.c_size_t), ("ncurr_pitch", C.c_size_t), ("vprev_pitch", C.c_size_t), ("nprev_pitch", C.c_size_t),
        ("rows", C.c_int), ("cols", C.c_int), ("iterations", C.c_int),
    ]


_lib = None


def load():
    """Load libyakhip.so; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libyakhip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C yak_amd/csrc`" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, sz, i, fp = C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_float)
    volp = C.POINTER(VolumeDesc)
    sigs = {
        "yakhip_ctx_create": [C.POINTER(vp), i],
        "yakhip_ctx_destroy": [vp],
        "yakhip_ctx_set_stream": [vp, vp],
        "yakhip_ctx_use_own_stream": [vp],
        "yakhip_ctx_synchronize": [vp],
        "yakhip_device_count": [C.POINTER(i)],
        "yakhip_device_name": [i, C.c_char_p, sz],
        "yakhip_ctx_set_option": [vp, C.c_char_p, i],
        "yakhip_ctx_stream": [vp, C.POINTER(C.c_void_p)],
        "yakhip_ctx_side_begin": [vp],
        "yakhip_ctx_side_end": [vp],
        "yakhip_ctx_kernel_time": [vp, C.POINTER(C.c_double), C.POINTER(i)],
        "yakhip_ctx_kernel_times": [vp, fp, fp, i, C.POINTER(i)],
        "yakhip_ctx_brick_stats": [vp, C.POINTER(C.c_ulonglong)],
        "yakhip_ctx_launch_counts": [vp, C.POINTER(C.c_ulonglong)],
        "yakhip_selftest_division": [vp, C.POINTER(C.c_ulonglong)],
        "yakhip_malloc": [vp, C.POINTER(vp), sz],
        "yakhip_malloc_pitch": [vp, C.POINTER(vp), C.POINTER(sz), sz, sz],
        "yakhip_free": [vp, vp],
        "yakhip_upload": [vp, vp, vp, sz],
        "yakhip_download": [vp, vp, vp, sz],
        "yakhip_upload_2d": [vp, vp, sz, vp, sz, sz, sz],
        "yakhip_download_2d": [vp, vp, sz, vp, sz, sz, sz],
        "yakhip_copy": [vp, vp, vp, sz],
        "yakhip_copy_2d": [vp, vp, sz, vp, sz, sz, sz],
        "yakhip_clear_volume": [vp, volp],
        "yakhip_surface_map_bytes": [volp, C.POINTER(sz)],
        "yakhip_surface_map_rebuild": [vp, volp],
        "yakhip_compute_dists": [vp, vp, sz, vp, sz, i, i, fp],
        "yakhip_integrate": [vp, volp, vp, sz, i, i, fp, fp, fp],
        "yakhip_integrate_static": [vp, volp, vp, sz, i, i, fp, fp, fp],
        "yakhip_integrate_prepare": [vp, volp, vp, sz, i, i],
        "yakhip_integrate_batch": [vp, volp, C.POINTER(FrameDesc), i],
        "yakhip_raycast_points": [vp, volp, fp, fp, fp, fp, C.c_float, C.c_float, vp, sz, vp, sz, i, i],
        "yakhip_raycast_depth": [vp, volp, fp, fp, fp, fp, C.c_float, C.c_float, vp, sz, vp, sz, i, i],
        "yakhip_raycast_points_pyramid": [vp, volp, fp, fp, fp, fp, C.c_float, C.c_float, vp, sz, vp, sz, i, i,
                                          vp, sz, vp, sz, vp, sz, vp, sz],
        "yakhip_raycast_slab": [vp, volp, i, i, fp, fp, fp, fp, C.c_float, C.c_float, vp, sz, vp, sz, vp, sz, vp, sz, i, i],
        "yakhip_raycast_halo": [volp, C.c_float, C.c_float, C.POINTER(i)],
        "yakhip_raycast_resolve": [vp, volp, i, i, fp, C.c_float, vp, sz, vp, sz, vp, sz, vp, sz, i, i],
        "yakhip_icp_accumulate_rows": [vp, fp, fp, fp, i, C.c_float, C.c_float, vp, sz, vp, sz, vp, sz, vp, sz, i, i, i, i,
                                       vp],
        "yakhip_bilateral_filter": [vp, vp, sz, vp, sz, i, i, i, C.c_float, C.c_float],
        "yakhip_truncate_depth": [vp, vp, sz, i, i, C.c_float],
        "yakhip_depth_pyramid": [vp, vp, sz, i, i, vp, sz, C.c_float],
        "yakhip_compute_points_normals": [vp, fp, vp, sz, i, i, vp, sz, vp, sz],
        "yakhip_resize_points_normals": [vp, vp, sz, vp, sz, i, i, vp, sz, vp, sz],
        "yakhip_resize_points_normals2": [vp, vp, sz, vp, sz, i, i, vp, sz, vp, sz, vp, sz, vp, sz],
        "yakhip_icp_estimate": [vp, fp, fp, fp, C.c_float, C.c_float, C.POINTER(IcpLevel), i, fp, fp, C.POINTER(i)],
        "yakhip_marching_cubes": [vp, volp, i, C.c_double, vp, C.c_ulonglong, C.POINTER(C.c_ulonglong)],
        "yakhip_marching_cubes_layers": [vp, volp, i, i, i, C.c_double, vp, C.c_ulonglong, C.POINTER(C.c_ulonglong)],
        "yakhip_weld_vertices": [vp, vp, C.c_ulonglong, vp, vp, C.POINTER(C.c_ulonglong)],
        "yakhip_extract_cloud": [vp, volp, fp, fp, vp, C.c_ulonglong, C.POINTER(C.c_ulonglong)],
        "yakhip_extract_normals": [vp, volp, vp, C.c_ulonglong, fp, fp, fp, C.c_float, vp],
        "yakhip_render_image_points": [vp, vp, sz, vp, sz, i, i, fp, vp, sz],
        "yakhip_render_image_depth": [vp, vp, sz, vp, sz, i, i, fp, fp, vp, sz],
        "yakhip_render_tangent_colors": [vp, vp, sz, i, i, vp, sz],
        "yakhip_icp_accumulate": [vp, fp, fp, fp, i, C.c_float, C.c_float, vp, sz, vp, sz, vp, sz, vp, sz, i, i, fp,
                                  C.POINTER(i)],
        "yakhip_comm_unique_id": [vp, sz],
        "yakhip_comm_create_rccl": [vp, C.POINTER(vp), vp, i, i],
        "yakhip_comm_create_local": [i, C.POINTER(vp)],
        "yakhip_comm_create_custom": [C.POINTER(vp), i, i, ALLREDUCE_FN, vp],
        "yakhip_comm_destroy": [vp],
        "yakhip_comm_abort": [vp],
        "yakhip_comm_info": [vp, C.POINTER(i), C.POINTER(i)],
        "yakhip_comm_allreduce": [vp, vp, vp, sz, i],
        "yakhip_comm_status": [vp, vp, C.POINTER(C.c_uint)],
        "yakhip_raycast_sharded": [vp, vp, volp, i, i, fp, fp, fp, fp, C.c_float, C.c_float, vp, sz, vp, sz, vp, sz, vp, sz,
                                   i, i],
        "yakhip_icp_estimate_sharded": [vp, vp, fp, fp, fp, C.c_float, C.c_float, C.POINTER(IcpLevel), i, fp, fp,
                                        C.POINTER(i)],
    }
    for name, argtypes in sigs.items():
        fn = getattr(L, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    L.yakhip_last_error.argtypes = []
    L.yakhip_last_error.restype = C.c_char_p
    _lib = L
    return L


def _chk(rc):
    if rc != 0:
        raise YakHipError(rc, load().yakhip_last_error().decode("utf-8", "replace"))


def _f32(values, n):
    vals = [float(v) for v in values]
    if len(vals) != n:
        raise ValueError("expected %d floats, got %d" % (n, len(vals)))
    return (C.c_float * n)(*vals)


def device_count():
    n = C.c_int(0)
    _chk(load().yakhip_device_count(C.byref(n)))
    return n.value


class Context:
    """One GPU, one stream: wraps yakhip_ctx.  By default work goes to torch's current stream so
    that torch tensors, torch events and these kernels are ordered on the same queue."""

    def __init__(self, device=0, use_torch_stream=True):
        if use_torch_stream:
            # torch first: it ships its own HIP runtime, and a process that has already initialised another
            # copy through libyakhip.so finds "no HIP GPUs" when torch initialises afterwards
            import torch
        L = load()
        self._h = C.c_void_p()
        _chk(L.yakhip_ctx_create(C.byref(self._h), int(device)))
        self.device = int(device)
        if use_torch_stream:
            self.set_stream(torch.cuda.current_stream(self.device).cuda_stream)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            load().yakhip_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, raw_stream):
        _chk(load().yakhip_ctx_set_stream(self._h, C.c_void_p(raw_stream or 0)))

    def set_option(self, key, value):
        _chk(load().yakhip_ctx_set_option(self._h, key.encode(), int(value)))

    def kernel_time(self):
        """(total device ms, launches) of the integrate kernel since kernel_timing was switched on."""
        ms, n = C.c_double(0.0), C.c_int(0)
        _chk(load().yakhip_ctx_kernel_time(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def kernel_times(self, capacity=4096):
        """Per-launch (device ms, start offset ms) lists of the bracketed launches since kernel_timing was set."""
        ms, st, n = (C.c_float * capacity)(), (C.c_float * capacity)(), C.c_int(0)
        _chk(load().yakhip_ctx_kernel_times(self._h, ms, st, capacity, C.byref(n)))
        k = min(n.value, capacity)
        return list(ms[:k]), list(st[:k])

    def brick_stats(self):
        """{dead, front, mixed} wave bricks of the integrate launches since the last call ("brick_stats" option)."""
        out = (C.c_ulonglong * 3)()
        _chk(load().yakhip_ctx_brick_stats(self._h, out))
        return {"dead": int(out[0]), "front": int(out[1]), "mixed": int(out[2])}

    def launch_counts(self):
        """{one_kernel, two_kernels}: how this context's yakhip_integrate launches ran so far ("defer_mixed" option)."""
        out = (C.c_ulonglong * 2)()
        _chk(load().yakhip_ctx_launch_counts(self._h, out))
        return {"one_kernel": int(out[0]), "two_kernels": int(out[1])}

    def selftest_division(self):
        out = (C.c_ulonglong * 4)()
        _chk(load().yakhip_selftest_division(self._h, out))
        return [int(v) for v in out]

    def synchronize(self):
        _chk(load().yakhip_ctx_synchronize(self._h))

    # ---- hot path; pointers are raw device addresses (ints) ----
    def stream_handle(self):
        """Raw hipStream_t (int) the context currently enqueues on."""
        h = C.c_void_p()
        _chk(load().yakhip_ctx_stream(self._h, C.byref(h)))
        return h.value or 0

    def side_begin(self):
        """Calls until side_end() go to the context's second stream (see include/yakhip.h)."""
        _chk(load().yakhip_ctx_side_begin(self._h))

    def side_end(self):
        _chk(load().yakhip_ctx_side_end(self._h))

    def clear_volume(self, vol):
        _chk(load().yakhip_clear_volume(self._h, C.byref(vol)))

    def surface_map_rebuild(self, vol):
        """vol.surface_map <- the bricks of the slab that hold a negative voxel (after voxels were written by other means)."""
        _chk(load().yakhip_surface_map_rebuild(self._h, C.byref(vol)))

    def compute_dists(self, depth_ptr, depth_pitch, dists_ptr, dists_pitch, rows, cols, intr):
        _chk(load().yakhip_compute_dists(self._h, C.c_void_p(depth_ptr), depth_pitch, C.c_void_p(dists_ptr),
                                         dists_pitch, rows, cols, _f32(intr, 4)))

    def integrate(self, vol, dists_ptr, dists_pitch, rows, cols, R, t, intr):
        _chk(load().yakhip_integrate(self._h, C.byref(vol), C.c_void_p(dists_ptr), dists_pitch, rows, cols,
                                     _f32(R, 9), _f32(t, 3), _f32(intr, 4)))

    def integrate_prepare(self, vol, dists_ptr, dists_pitch, rows, cols):
        """The pose-independent part of integrate(), ahead of time on the current stream (include/yakhip.h)."""
        _chk(load().yakhip_integrate_prepare(self._h, C.byref(vol), C.c_void_p(dists_ptr), dists_pitch, rows, cols))

    def integrate_static(self, vol, dists_ptr, dists_pitch, rows, cols, R, t, intr):
        """yakhip_integrate without any allocation in the call (scratch in the library's file-scope array)."""
        _chk(load().yakhip_integrate_static(self._h, C.byref(vol), C.c_void_p(dists_ptr), dists_pitch, rows, cols,
                                            _f32(R, 9), _f32(t, 3), _f32(intr, 4)))

    def integrate_batch(self, vol, frames):
        """frames: sequence of (dists_ptr, dists_pitch, rows, cols, R, t, intr), applied in order."""
        arr = (FrameDesc * len(frames))()
        for d, (ptr, pitch, rows, cols, R, t, intr) in zip(arr, frames):
            d.dists, d.dists_pitch, d.rows, d.cols = ptr, pitch, rows, cols
            d.R[:] = [float(v) for v in R]
            d.t[:] = [float(v) for v in t]
            d.intr[:] = [float(v) for v in intr]
        _chk(load().yakhip_integrate_batch(self._h, C.byref(vol), arr, len(frames)))

    # ---- host <-> device copies (blocking; the reference's DeviceMemory[2D]::upload / download) ----
    def upload(self, dst_ptr, src, nbytes):
        _chk(load().yakhip_upload(self._h, C.c_void_p(dst_ptr), C.c_void_p(src), nbytes))

    def download(self, dst, src_ptr, nbytes):
        _chk(load().yakhip_download(self._h, C.c_void_p(dst), C.c_void_p(src_ptr), nbytes))

    def upload_2d(self, dst_ptr, dst_pitch, src, src_pitch, width_bytes, rows):
        _chk(load().yakhip_upload_2d(self._h, C.c_void_p(dst_ptr), dst_pitch, C.c_void_p(src), src_pitch, width_bytes, rows))

    def download_2d(self, dst, dst_pitch, src_ptr, src_pitch, width_bytes, rows):
        _chk(load().yakhip_download_2d(self._h, C.c_void_p(dst), dst_pitch, C.c_void_p(src_ptr), src_pitch, width_bytes,
                                       rows))

    def copy(self, dst_ptr, src_ptr, nbytes):
        _chk(load().yakhip_copy(self._h, C.c_void_p(dst_ptr), C.c_void_p(src_ptr), nbytes))

    def copy_2d(self, dst_ptr, dst_pitch, src_ptr, src_pitch, width_bytes, rows):
        _chk(load().yakhip_copy_2d(self._h, C.c_void_p(dst_ptr), dst_pitch, C.c_void_p(src_ptr), src_pitch, width_bytes, rows))

    def raycast_points(self, vol, R, t, Rinv, intr, points_ptr, points_pitch, normals_ptr, normals_pitch, rows, cols,
                       step_factor=0.75, gradient_delta_factor=0.5):
        _chk(load().yakhip_raycast_points(self._h, C.byref(vol), _f32(R, 9), _f32(t, 3), _f32(Rinv, 9), _f32(intr, 4),
                                          step_factor, gradient_delta_factor, C.c_void_p(points_ptr), points_pitch,
                                          C.c_void_p(normals_ptr), normals_pitch, rows, cols))

    def raycast_points_pyramid(self, vol, R, t, Rinv, intr, points_ptr, points_pitch, normals_ptr, normals_pitch, rows, cols,
                               half, quarter=None, step_factor=0.75, gradient_delta_factor=0.5):
        """raycast_points plus the half- and (optionally) quarter-size maps from the same launch; half / quarter =
        (points_ptr, points_pitch, normals_ptr, normals_pitch)."""
        q = quarter if quarter is not None else (None, 0, None, 0)
        _chk(load().yakhip_raycast_points_pyramid(self._h, C.byref(vol), _f32(R, 9), _f32(t, 3), _f32(Rinv, 9), _f32(intr, 4),
                                                  step_factor, gradient_delta_factor, C.c_void_p(points_ptr), points_pitch,
                                                  C.c_void_p(normals_ptr), normals_pitch, rows, cols,
                                                  C.c_void_p(half[0]), half[1], C.c_void_p(half[2]), half[3],
                                                  C.c_void_p(q[0]), q[1], C.c_void_p(q[2]), q[3]))

    def raycast_depth(self, vol, R, t, Rinv, intr, depth_ptr, depth_pitch, normals_ptr, normals_pitch, rows, cols,
                      step_factor=0.75, gradient_delta_factor=0.5):
        _chk(load().yakhip_raycast_depth(self._h, C.byref(vol), _f32(R, 9), _f32(t, 3), _f32(Rinv, 9), _f32(intr, 4),
                                         step_factor, gradient_delta_factor, C.c_void_p(depth_ptr), depth_pitch,
                                         C.c_void_p(normals_ptr), normals_pitch, rows, cols))

    def raycast_slab(self, vol, z_own0, z_own1, R, t, Rinv, intr, keys_ptr, keys_pitch, normals_ptr, normals_pitch, rows,
                     cols, points_ptr=None, points_pitch=0, depth_ptr=None, depth_pitch=0, step_factor=0.75,
                     gradient_delta_factor=0.5):
        """vol = the layers HELD (own + halo); [z_own0, z_own1) = the slab's own.  See include/yakhip.h."""
        _chk(load().yakhip_raycast_slab(self._h, C.byref(vol), int(z_own0), int(z_own1), _f32(R, 9), _f32(t, 3),
                                        _f32(Rinv, 9), _f32(intr, 4), step_factor, gradient_delta_factor,
                                        C.c_void_p(points_ptr), points_pitch, C.c_void_p(depth_ptr), depth_pitch,
                                        C.c_void_p(normals_ptr), normals_pitch, C.c_void_p(keys_ptr), keys_pitch, rows,
                                        cols))

    def raycast_resolve(self, vol, z_own0, z_own1, Rinv, keys_ptr, keys_pitch, normals_ptr, normals_pitch, rows, cols,
                        points_ptr=None, points_pitch=0, depth_ptr=None, depth_pitch=0, gradient_delta_factor=0.5):
        """Second step of the slab composite, on the composited images.  See include/yakhip.h."""
        _chk(load().yakhip_raycast_resolve(self._h, C.byref(vol), int(z_own0), int(z_own1), _f32(Rinv, 9),
                                           gradient_delta_factor, C.c_void_p(points_ptr), points_pitch,
                                           C.c_void_p(depth_ptr), depth_pitch, C.c_void_p(normals_ptr), normals_pitch,
                                           C.c_void_p(keys_ptr), keys_pitch, rows, cols))

    # ---- tracking side ----
    def bilateral_filter(self, src_ptr, src_pitch, dst_ptr, dst_pitch, rows, cols, kernel_size, sigma_spatial,
                         sigma_depth):
        _chk(load().yakhip_bilateral_filter(self._h, C.c_void_p(src_ptr), src_pitch, C.c_void_p(dst_ptr), dst_pitch,
                                            rows, cols, kernel_size, sigma_spatial, sigma_depth))

    def truncate_depth(self, depth_ptr, pitch, rows, cols, max_dist):
        _chk(load().yakhip_truncate_depth(self._h, C.c_void_p(depth_ptr), pitch, rows, cols, max_dist))

    def depth_pyramid(self, src_ptr, src_pitch, src_rows, src_cols, dst_ptr, dst_pitch, sigma_depth):
        _chk(load().yakhip_depth_pyramid(self._h, C.c_void_p(src_ptr), src_pitch, src_rows, src_cols,
                                         C.c_void_p(dst_ptr), dst_pitch, sigma_depth))

    def compute_points_normals(self, intr, depth_ptr, depth_pitch, rows, cols, points_ptr, points_pitch, normals_ptr,
                               normals_pitch):
        _chk(load().yakhip_compute_points_normals(self._h, _f32(intr, 4), C.c_void_p(depth_ptr), depth_pitch, rows, cols,
                                                  C.c_void_p(points_ptr), points_pitch, C.c_void_p(normals_ptr),
                                                  normals_pitch))

    def resize_points_normals(self, points_ptr, points_pitch, normals_ptr, normals_pitch, src_rows, src_cols,
                              points_out_ptr, points_out_pitch, normals_out_ptr, normals_out_pitch):
        _chk(load().yakhip_resize_points_normals(self._h, C.c_void_p(points_ptr), points_pitch, C.c_void_p(normals_ptr),
                                                 normals_pitch, src_rows, src_cols, C.c_void_p(points_out_ptr),
                                                 points_out_pitch, C.c_void_p(normals_out_ptr), normals_out_pitch))

    def resize_points_normals2(self, points_ptr, points_pitch, normals_ptr, normals_pitch, src_rows, src_cols,
                               points1_ptr, points1_pitch, normals1_ptr, normals1_pitch,
                               points2_ptr, points2_pitch, normals2_ptr, normals2_pitch):
        """Half- and quarter-size maps in one launch (the same bits as two resize_points_normals calls)."""
        _chk(load().yakhip_resize_points_normals2(self._h, C.c_void_p(points_ptr), points_pitch, C.c_void_p(normals_ptr),
                                                  normals_pitch, src_rows, src_cols, C.c_void_p(points1_ptr), points1_pitch,
                                                  C.c_void_p(normals1_ptr), normals1_pitch, C.c_void_p(points2_ptr),
                                                  points2_pitch, C.c_void_p(normals2_ptr), normals2_pitch))

    def icp_accumulate(self, R, t, intr, level, dist_thres, angle_thres, vcurr_ptr, vcurr_pitch, ncurr_ptr, ncurr_pitch,
                       vprev_ptr, vprev_pitch, nprev_ptr, nprev_pitch, rows, cols):
        """Blocking.  Returns (sums: list of 27 floats, counts: list of 6 ints)."""
        sums, counts = (C.c_float * 27)(), (C.c_int * 6)()
        _chk(load().yakhip_icp_accumulate(self._h, _f32(R, 9), _f32(t, 3), _f32(intr, 4), int(level), dist_thres,
                                          angle_thres, C.c_void_p(vcurr_ptr), vcurr_pitch, C.c_void_p(ncurr_ptr),
                                          ncurr_pitch, C.c_void_p(vprev_ptr), vprev_pitch, C.c_void_p(nprev_ptr),
                                          nprev_pitch, rows, cols, sums, counts))
        return list(sums), list(counts)

    def icp_accumulate_rows(self, R, t, intr, level, dist_thres, angle_thres, vcurr_ptr, vcurr_pitch, ncurr_ptr,
                            ncurr_pitch, vprev_ptr, vprev_pitch, nprev_ptr, nprev_pitch, rows, cols, row_begin, row_end,
                            sums_dev_ptr):
        """Not blocking: 33 floats (27 sums, 6 counts) for rows [row_begin, row_end) go to device memory."""
        _chk(load().yakhip_icp_accumulate_rows(self._h, _f32(R, 9), _f32(t, 3), _f32(intr, 4), int(level), dist_thres,
                                               angle_thres, C.c_void_p(vcurr_ptr), vcurr_pitch, C.c_void_p(ncurr_ptr),
                                               ncurr_pitch, C.c_void_p(vprev_ptr), vprev_pitch, C.c_void_p(nprev_ptr),
                                               nprev_pitch, rows, cols, int(row_begin), int(row_end),
                                               C.c_void_p(sums_dev_ptr)))

    def icp_estimate(self, R, t, intr, dist_thres, angle_thres, levels):
        """levels: per pyramid level l = 0.. a tuple (vcurr_ptr, ncurr_ptr, vprev_ptr, nprev_ptr, pitch, rows, cols,
        iterations).  Blocking.  Returns (R 3x3 list, t list, ok)."""
        arr = (IcpLevel * len(levels))()
        for d, (vc, nc, vp_, np_, pitch, rows, cols, iters) in zip(arr, levels):
            d.vcurr, d.ncurr, d.vprev, d.nprev = vc, nc, vp_, np_
            d.vcurr_pitch = d.ncurr_pitch = d.vprev_pitch = d.nprev_pitch = pitch
            d.rows, d.cols, d.iterations = rows, cols, iters
        Ro, to, ok = (C.c_float * 9)(), (C.c_float * 3)(), C.c_int(0)
        _chk(load().yakhip_icp_estimate(self._h, _f32(R, 9), _f32(t, 3), _f32(intr, 4), dist_thres, angle_thres, arr,
                                        len(levels), Ro, to, C.byref(ok)))
        return list(Ro), list(to), bool(ok.value)

    def marching_cubes(self, vol, min_weight, scale, triangles_ptr=0, capacity=0):
        """Blocking.  Returns the number of triangles the slab produces (written iff it fits `capacity`)."""
        n = C.c_ulonglong(0)
        _chk(load().yakhip_marching_cubes(self._h, C.byref(vol), int(min_weight), float(scale),
                                          C.c_void_p(triangles_ptr or 0), int(capacity), C.byref(n)))
        return n.value

    def marching_cubes_layers(self, vol, z_first, z_last, min_weight, scale, triangles_ptr=0, capacity=0):
        """marching_cubes over the layers [z_first, z_last) of the slab `vol` (whose surface map still applies)."""
        n = C.c_ulonglong(0)
        _chk(load().yakhip_marching_cubes_layers(self._h, C.byref(vol), int(z_first), int(z_last), int(min_weight), float(scale),
                                                 C.c_void_p(triangles_ptr or 0), int(capacity), C.byref(n)))
        return n.value

    def weld_vertices(self, triangles_ptr, n_triangles, vertices_ptr, index_ptr):
        """Blocking.  Merges bit-identical corners; returns the number of distinct vertices written."""
        n = C.c_ulonglong(0)
        _chk(load().yakhip_weld_vertices(self._h, C.c_void_p(triangles_ptr or 0), int(n_triangles),
                                         C.c_void_p(vertices_ptr or 0), C.c_void_p(index_ptr or 0), C.byref(n)))
        return n.value

    def extract_cloud(self, vol, R, t, points_ptr=None, capacity=0):
        """Blocking.  Returns the number of points the slab holds; writes min(n, capacity) of them."""
        n = C.c_ulonglong(0)
        _chk(load().yakhip_extract_cloud(self._h, C.byref(vol), _f32(R, 9), _f32(t, 3), C.c_void_p(points_ptr), int(capacity),
                                         C.byref(n)))
        return n.value

    def extract_normals(self, vol, points_ptr, n_points, R, t, Rinv, normals_ptr, gradient_delta_factor=0.5):
        _chk(load().yakhip_extract_normals(self._h, C.byref(vol), C.c_void_p(points_ptr), int(n_points), _f32(R, 9), _f32(t, 3),
                                           _f32(Rinv, 9), gradient_delta_factor, C.c_void_p(normals_ptr)))

    # ---- sharded volume: the exchange steps (include/yakhip.h, yakhip_comm) ----
    def allreduce(self, comm, buf_ptr, count, op):
        _chk(load().yakhip_comm_allreduce(self._h, comm._h, C.c_void_p(buf_ptr), int(count), int(op)))

    def raycast_sharded(self, comm, vol, z_own0, z_own1, R, t, Rinv, intr, keys_ptr, keys_pitch, normals_ptr, normals_pitch,
                        rows, cols, points_ptr=None, points_pitch=0, depth_ptr=None, depth_pitch=0, step_factor=0.75,
                        gradient_delta_factor=0.5):
        """Slab ray cast + composite over the ranks + resolve + composite of the settled pixels; not waited for."""
        _chk(load().yakhip_raycast_sharded(self._h, comm._h, C.byref(vol), int(z_own0), int(z_own1), _f32(R, 9), _f32(t, 3),
                                           _f32(Rinv, 9), _f32(intr, 4), step_factor, gradient_delta_factor,
                                           C.c_void_p(points_ptr), points_pitch, C.c_void_p(depth_ptr), depth_pitch,
                                           C.c_void_p(normals_ptr), normals_pitch, C.c_void_p(keys_ptr), keys_pitch, rows,
                                           cols))

    def comm_status(self, comm):
        """Blocking.  (pixels settled by the last sharded ray cast's second step, calls whose list overflowed)."""
        out = (C.c_uint * 2)()
        _chk(load().yakhip_comm_status(self._h, comm._h, out))
        return int(out[0]), int(out[1])

    def icp_estimate_sharded(self, comm, R, t, intr, dist_thres, angle_thres, levels):
        """icp_estimate over the ranks of `comm` (row bands, all-reduced sums).  Blocking.  Returns (R, t, ok)."""
        arr = (IcpLevel * len(levels))()
        for d, (vc, nc, vp_, np_, pitch, rows, cols, iters) in zip(arr, levels):
            d.vcurr, d.ncurr, d.vprev, d.nprev = vc, nc, vp_, np_
            d.vcurr_pitch = d.ncurr_pitch = d.vprev_pitch = d.nprev_pitch = pitch
            d.rows, d.cols, d.iterations = rows, cols, iters
        Ro, to, ok = (C.c_float * 9)(), (C.c_float * 3)(), C.c_int(0)
        _chk(load().yakhip_icp_estimate_sharded(self._h, comm._h, _f32(R, 9), _f32(t, 3), _f32(intr, 4), dist_thres,
                                                angle_thres, arr, len(levels), Ro, to, C.byref(ok)))
        return list(Ro), list(to), bool(ok.value)

    # ---- debug display: images are 4 bytes per pixel {b, g, r, 0} ----
    def render_image_points(self, points_ptr, points_pitch, normals_ptr, normals_pitch, rows, cols, light_pose, image_ptr,
                            image_pitch):
        _chk(load().yakhip_render_image_points(self._h, C.c_void_p(points_ptr), points_pitch, C.c_void_p(normals_ptr),
                                               normals_pitch, rows, cols, _f32(light_pose, 3), C.c_void_p(image_ptr),
                                               image_pitch))

    def render_image_depth(self, depth_ptr, depth_pitch, normals_ptr, normals_pitch, rows, cols, intr, light_pose,
                           image_ptr, image_pitch):
        _chk(load().yakhip_render_image_depth(self._h, C.c_void_p(depth_ptr), depth_pitch, C.c_void_p(normals_ptr),
                                              normals_pitch, rows, cols, _f32(intr, 4), _f32(light_pose, 3),
                                              C.c_void_p(image_ptr), image_pitch))

    def render_tangent_colors(self, normals_ptr, normals_pitch, rows, cols, image_ptr, image_pitch):
        _chk(load().yakhip_render_tangent_colors(self._h, C.c_void_p(normals_ptr), normals_pitch, rows, cols,
                                                 C.c_void_p(image_ptr), image_pitch))


class Comm:
    """yakhip_comm: the ranks that share a z-slab sharded volume."""

    def __init__(self, handle, keep=None):
        self._h, self._keep = handle, keep

    @staticmethod
    def unique_id():
        buf = (C.c_char * 128)()
        _chk(load().yakhip_comm_unique_id(buf, 128))
        return bytes(buf)

    @classmethod
    def rccl(cls, ctx, unique_id, world, rank):
        h = C.c_void_p()
        _chk(load().yakhip_comm_create_rccl(ctx._h, C.byref(h), C.c_char_p(unique_id), int(world), int(rank)))
        return cls(h)

    @classmethod
    def local(cls, world):
        """`world` communicators for ranks that are threads of this process."""
        arr = (C.c_void_p * world)()
        _chk(load().yakhip_comm_create_local(int(world), arr))
        return [cls(C.c_void_p(arr[r])) for r in range(world)]

    @classmethod
    def custom(cls, world, rank, fn):
        """fn(buf_ptr, count, op, stream_ptr) -> 0: the caller's all-reduce (see include/yakhip.h)."""
        cb = ALLREDUCE_FN(lambda user, buf, count, op, stream: int(fn(buf, count, op, stream) or 0))
        h = C.c_void_p()
        _chk(load().yakhip_comm_create_custom(C.byref(h), int(world), int(rank), cb, None))
        return cls(h, keep=cb)

    def info(self):
        r, w = C.c_int(0), C.c_int(0)
        _chk(load().yakhip_comm_info(self._h, C.byref(r), C.byref(w)))
        return r.value, w.value

    def abort(self):
        """yakhip_comm_abort: peers waiting on this rank return an error; every later collective here fails."""
        _chk(load().yakhip_comm_abort(self._h))

    def close(self):
        if self._h is not None and self._h.value:
            load().yakhip_comm_destroy(self._h)
            self._h = C.c_void_p()


def raycast_halo(vol, step_factor=0.75, gradient_delta_factor=0.5):
    """Halo layers a slab needs on either side for yakhip_raycast_slab."""
    n = C.c_int(0)
    _chk(load().yakhip_raycast_halo(C.byref(vol), step_factor, gradient_delta_factor, C.byref(n)))
    return n.value


def surface_map_bytes(vol):
    """Bytes of the surface map of the slab `vol` (one bit per 16^3 brick, yakhip.h)."""
    n = C.c_size_t(0)
    _chk(load().yakhip_surface_map_bytes(C.byref(vol), C.byref(n)))
    return n.value


def volume_desc(data_ptr, dims, voxel_size, trunc_dist, max_weight, z0=0, z1=None, surface_map=None):
    """surface_map: device pointer of surface_map_bytes() bytes, or None (ray casts then fetch every sample)."""
    v = VolumeDesc()
    v.surface_map = surface_map
    v.data = data_ptr
    v.dims[:] = [int(d) for d in dims]
    v.z0 = int(z0)
    v.z1 = int(dims[2] if z1 is None else z1)
    v.voxel_size[:] = [float(s) for s in voxel_size]
    v.trunc_dist = float(trunc_dist)
    v.max_weight = int(max_weight)
    return v
