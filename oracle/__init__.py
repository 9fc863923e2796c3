"""ctypes view of the CPU oracle (oracle/liboracle.so) and of oracle/_ref/libyakref.so.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package `yak_amd`.
See oracle/yak_oracle.h for the parity status ("parity unpinned" for integrate).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
_REF = os.path.join(_HERE, "_ref", "libyakref.so")


def build(force=False):
    """Compile liboracle.so (and _ref when /root/reference is present)."""
    if force or not os.path.exists(_LIB) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB)
        for f in os.listdir(_HERE)
        if f.endswith((".c", ".h"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"])
    if os.path.isdir("/root/reference/yak"):
        subprocess.check_call(["make", "-C", _HERE, "ref"])


_lib = None
_ref = None

_u16p = np.ctypeslib.ndpointer(np.uint16, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = C.CDLL(_LIB)
        L.yko_float2half_rn.restype = C.c_uint16
        L.yko_float2half_rn.argtypes = [C.c_float]
        L.yko_half2float.restype = C.c_float
        L.yko_half2float.argtypes = [C.c_uint16]
        L.yko_float2half_array.restype = None
        L.yko_float2half_array.argtypes = [_f32p, _u16p, C.c_size_t]
        L.yko_half2float_array.restype = None
        L.yko_half2float_array.argtypes = [_u16p, _f32p, C.c_size_t]
        L.yko_pack_tsdf.restype = C.c_uint32
        L.yko_pack_tsdf.argtypes = [C.c_float, C.c_int]
        L.yko_unpack_tsdf.restype = C.c_float
        L.yko_unpack_tsdf.argtypes = [C.c_uint32, C.POINTER(C.c_int)]
        L.yko_compute_dists.restype = None
        L.yko_compute_dists.argtypes = [_u16p, C.c_size_t, _u16p, C.c_size_t, C.c_int, C.c_int,
                                        C.c_float, C.c_float, C.c_float, C.c_float]
        L.yko_clear_volume.restype = None
        L.yko_clear_volume.argtypes = [_u32p, C.c_size_t]
        L.yko_integrate.restype = None
        L.yko_integrate.argtypes = [_u32p, _i32p, C.c_int, C.c_int, _f32p, C.c_float, C.c_int,
                                    _u16p, C.c_size_t, C.c_int, C.c_int, _f32p, _f32p, _f32p, _u64p, C.c_int]
        L.yko_raycast.restype = None
        L.yko_raycast.argtypes = [_u32p, _i32p, _f32p, C.c_float, _f32p, _f32p, _f32p, _f32p, C.c_float, C.c_float,
                                  C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                                  C.c_size_t, _u64p, C.c_int]
        L.yko_raycast_slab.restype = None
        L.yko_raycast_slab.argtypes = L.yko_raycast.argtypes + [C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        vp, sz, i, f = C.c_void_p, C.c_size_t, C.c_int, C.c_float
        L.yko_bilateral_filter.restype = None
        L.yko_bilateral_filter.argtypes = [vp, sz, vp, sz, i, i, i, f, f]
        L.yko_truncate_depth.restype = None
        L.yko_truncate_depth.argtypes = [vp, sz, i, i, f]
        L.yko_depth_pyramid.restype = None
        L.yko_depth_pyramid.argtypes = [vp, sz, i, i, vp, sz, f]
        L.yko_compute_points_normals.restype = None
        L.yko_compute_points_normals.argtypes = [vp, sz, i, i, _f32p, vp, sz, vp, sz]
        L.yko_resize_points_normals.restype = None
        L.yko_resize_points_normals.argtypes = [vp, sz, vp, sz, i, i, vp, sz, vp, sz]
        L.yko_icp_accumulate.restype = None
        L.yko_icp_accumulate.argtypes = [_f32p, _f32p, _f32p, i, f, f, vp, sz, vp, sz, vp, sz, vp, sz, i, i,
                                         np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS"), _u64p, vp]
        L.yko_marching_cubes.restype = C.c_uint64
        L.yko_marching_cubes.argtypes = [_u32p, _i32p, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        L.yko_marching_cubes_mt.restype = C.c_uint64
        L.yko_marching_cubes_mt.argtypes = [_u32p, _i32p, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_uint64, C.c_int]
        L.yko_max_threads.restype = C.c_int
        _lib = L
    return _lib


def ref():
    """The reference's stand-alone-compilable pieces; None when oracle/_ref was never built."""
    global _ref
    if _ref is None:
        if not os.path.exists(_REF):
            return None
        R = C.CDLL(_REF)
        R.yakref_half2float.restype = C.c_float
        R.yakref_half2float.argtypes = [C.c_uint16]
        R.yakref_float2half_ctor.restype = C.c_uint16
        R.yakref_float2half_ctor.argtypes = [C.c_float]
        R.yakref_pack.restype = C.c_uint32
        R.yakref_pack.argtypes = [C.c_float, C.c_uint16]
        R.yakref_unpack.restype = None
        R.yakref_unpack.argtypes = [C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_uint16)]
        for name in ("yakref_edge_flags", "yakref_triangle_table", "yakref_edge_connections"):
            getattr(R, name).restype = C.POINTER(C.c_int)
        _ref = R
    return _ref


# ----------------------------------------------------------------------------------------------
def float2half(values):
    v = np.ascontiguousarray(values, dtype=np.float32).ravel()
    out = np.empty(v.size, dtype=np.uint16)
    lib().yko_float2half_array(v, out, v.size)
    return out.reshape(np.shape(values))


def half2float(bits):
    b = np.ascontiguousarray(bits, dtype=np.uint16).ravel()
    out = np.empty(b.size, dtype=np.float32)
    lib().yko_half2float_array(b, out, b.size)
    return out.reshape(np.shape(bits))


def compute_dists(depth, intr):
    """depth: (rows, cols) uint16 mm -> (rows, cols) uint16 half bits (ray length, metres)."""
    depth = np.ascontiguousarray(depth, dtype=np.uint16)
    rows, cols = depth.shape
    out = np.empty_like(depth)
    fx, fy, cx, cy = (float(v) for v in intr)
    lib().yko_compute_dists(depth, cols * 2, out, cols * 2, rows, cols, fx, fy, cx, cy)
    return out


def integrate(slab, dims, voxel_size, trunc_dist, max_weight, dists, R, t, intr, z0=0, z1=None, nthreads=1):
    """In-place update of `slab` ((z1-z0)*Y*X uint32, C order z,y,x).  Returns (visited, updated)."""
    dims = np.ascontiguousarray(dims, dtype=np.int32)
    if z1 is None:
        z1 = int(dims[2])
    assert slab.dtype == np.uint32 and slab.flags.c_contiguous
    assert slab.size == (z1 - z0) * int(dims[0]) * int(dims[1])
    dists = np.ascontiguousarray(dists, dtype=np.uint16)
    rows, cols = dists.shape
    stats = np.zeros(2, dtype=np.uint64)
    lib().yko_integrate(
        slab.reshape(-1), dims, int(z0), int(z1), np.ascontiguousarray(voxel_size, dtype=np.float32),
        float(trunc_dist), int(max_weight), dists, cols * 2, rows, cols,
        np.ascontiguousarray(R, dtype=np.float32).reshape(-1), np.ascontiguousarray(t, dtype=np.float32),
        np.ascontiguousarray(intr, dtype=np.float32), stats, int(nthreads))
    return int(stats[0]), int(stats[1])


def raycast(volume, dims, voxel_size, trunc_dist, R, t, Rinv, intr, rows, cols, step_factor=0.75,
            gradient_delta_factor=0.5, want_depth=False, nthreads=1):
    """Ray-cast the whole volume from cam2vol = (R, t).  Returns (points|depth, normals, fetches, hits):
    points/normals are (rows, cols, 4) float32 (NaN where no hit), depth is (rows, cols) uint16 mm."""
    dims = np.ascontiguousarray(dims, dtype=np.int32)
    assert volume.dtype == np.uint32 and volume.flags.c_contiguous
    assert volume.size == int(dims[0]) * int(dims[1]) * int(dims[2])
    normals = np.empty((rows, cols, 4), dtype=np.float32)
    points = None if want_depth else np.empty((rows, cols, 4), dtype=np.float32)
    depth = np.empty((rows, cols), dtype=np.uint16) if want_depth else None
    stats = np.zeros(2, dtype=np.uint64)
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32).reshape(-1)  # noqa: E731
    lib().yko_raycast(
        volume.reshape(-1), dims, f32(voxel_size), float(trunc_dist), f32(R), f32(t), f32(Rinv), f32(intr),
        float(step_factor), float(gradient_delta_factor), int(rows), int(cols),
        points.ctypes.data if points is not None else None, cols * 16, normals.ctypes.data, cols * 16,
        depth.ctypes.data if depth is not None else None, cols * 2, stats, int(nthreads))
    return (depth if want_depth else points), normals, int(stats[0]), int(stats[1])


def raycast_slab(volume, dims, voxel_size, trunc_dist, R, t, Rinv, intr, rows, cols, z_own0, z_own1, step_factor=0.75,
                 gradient_delta_factor=0.5, want_depth=False, nthreads=1):
    """The march of `raycast` restricted to the sample pairs whose first sample lies in the layers
    [z_own0, z_own1) (the whole volume is still passed).  Returns (points|depth, normals, keys): keys is
    (rows, cols) int32, 2 * index of the march-ending pair or INT32_MAX."""
    dims = np.ascontiguousarray(dims, dtype=np.int32)
    assert volume.dtype == np.uint32 and volume.flags.c_contiguous
    assert volume.size == int(dims[0]) * int(dims[1]) * int(dims[2])
    normals = np.empty((rows, cols, 4), dtype=np.float32)
    points = None if want_depth else np.empty((rows, cols, 4), dtype=np.float32)
    depth = np.empty((rows, cols), dtype=np.uint16) if want_depth else None
    keys = np.empty((rows, cols), dtype=np.int32)
    stats = np.zeros(2, dtype=np.uint64)
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32).reshape(-1)  # noqa: E731
    lib().yko_raycast_slab(
        volume.reshape(-1), dims, f32(voxel_size), float(trunc_dist), f32(R), f32(t), f32(Rinv), f32(intr),
        float(step_factor), float(gradient_delta_factor), int(rows), int(cols),
        points.ctypes.data if points is not None else None, cols * 16, normals.ctypes.data, cols * 16,
        depth.ctypes.data if depth is not None else None, cols * 2, stats, int(nthreads), int(z_own0), int(z_own1),
        keys.ctypes.data, cols * 4)
    return (depth if want_depth else points), normals, keys


def _u16img(a):
    a = np.ascontiguousarray(a, dtype=np.uint16)
    assert a.ndim == 2
    return a


def _f4img(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 3 and a.shape[2] == 4
    return a


def bilateral_filter(depth, kernel_size=7, sigma_spatial=4.5, sigma_depth=0.04):
    src = _u16img(depth)
    rows, cols = src.shape
    dst = np.empty_like(src)
    lib().yko_bilateral_filter(src.ctypes.data, cols * 2, dst.ctypes.data, cols * 2, rows, cols, int(kernel_size),
                               float(sigma_spatial), float(sigma_depth))
    return dst


def truncate_depth(depth, max_dist):
    out = _u16img(depth).copy()
    rows, cols = out.shape
    lib().yko_truncate_depth(out.ctypes.data, cols * 2, rows, cols, float(max_dist))
    return out


def depth_pyramid(depth, sigma_depth=0.04):
    src = _u16img(depth)
    rows, cols = src.shape
    dst = np.empty((rows // 2, cols // 2), dtype=np.uint16)
    lib().yko_depth_pyramid(src.ctypes.data, cols * 2, rows, cols, dst.ctypes.data, (cols // 2) * 2, float(sigma_depth))
    return dst


def compute_points_normals(depth, intr):
    """intr = the intrinsics of this pyramid level.  Returns (points, normals), each (rows, cols, 4) float32."""
    src = _u16img(depth)
    rows, cols = src.shape
    pts = np.empty((rows, cols, 4), dtype=np.float32)
    nrm = np.empty_like(pts)
    lib().yko_compute_points_normals(src.ctypes.data, cols * 2, rows, cols,
                                     np.ascontiguousarray(intr, dtype=np.float32), pts.ctypes.data, cols * 16,
                                     nrm.ctypes.data, cols * 16)
    return pts, nrm


def resize_points_normals(points, normals):
    p, n = _f4img(points), _f4img(normals)
    rows, cols = p.shape[:2]
    po = np.empty((rows // 2, cols // 2, 4), dtype=np.float32)
    no = np.empty_like(po)
    lib().yko_resize_points_normals(p.ctypes.data, cols * 16, n.ctypes.data, cols * 16, rows, cols, po.ctypes.data,
                                    (cols // 2) * 16, no.ctypes.data, (cols // 2) * 16)
    return po, no


def icp_accumulate(R, t, intr, level, dist_thres, angle_thres, vcurr, ncurr, vprev, nprev, want_codes=False):
    """Returns (sums float64[27], counts uint64[6][, codes uint8 (rows, cols)])."""
    vc, nc, vp_, np_ = _f4img(vcurr), _f4img(ncurr), _f4img(vprev), _f4img(nprev)
    rows, cols = vc.shape[:2]
    assert nc.shape == vc.shape == vp_.shape == np_.shape
    sums = np.zeros(27, dtype=np.float64)
    counts = np.zeros(6, dtype=np.uint64)
    codes = np.zeros((rows, cols), dtype=np.uint8) if want_codes else None
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32).reshape(-1)  # noqa: E731
    lib().yko_icp_accumulate(f32(R), f32(t), f32(intr), int(level), float(dist_thres), float(angle_thres),
                             vc.ctypes.data, cols * 16, nc.ctypes.data, cols * 16, vp_.ctypes.data, cols * 16,
                             np_.ctypes.data, cols * 16, rows, cols, sums, counts,
                             codes.ctypes.data if codes is not None else None)
    return (sums, counts, codes) if want_codes else (sums, counts)


def mc_tables():
    """(tri_table int32[256,16], edge_connections int32[12,2], source): the reference's own compiled
    arrays when oracle/_ref is built, otherwise the product's packed table unpacked."""
    R = ref()
    if R is not None:
        tri = np.ctypeslib.as_array(R.yakref_triangle_table(), shape=(256 * 16,)).astype(np.int32).reshape(256, 16)
        conn = np.ctypeslib.as_array(R.yakref_edge_connections(), shape=(24,)).astype(np.int32).reshape(12, 2)
        return tri, conn, "reference"
    tri, conn = unpack_case_table()
    return tri, conn, "packed"


def unpack_case_table():
    """yak_amd/csrc/mc_case_table.h parsed back into the classic 256 x 16 / 12 x 2 integer arrays."""
    import re

    text = open(os.path.join(os.path.dirname(_HERE), "yak_amd", "csrc", "mc_case_table.h")).read()
    body = text[text.index("yak_mc_case_table[256]"):]
    words = [int(w, 16) for w in re.findall(r"0x([0-9a-f]{16})ull", body)]
    assert len(words) == 256
    tri = np.full((256, 16), -1, dtype=np.int32)
    for c, w in enumerate(words):
        for k in range(3 * (w >> 60)):
            tri[c, k] = (w >> (4 * k)) & 0xF
    pairs = re.findall(r"\{ (\d+), (\d+) \}", text[text.index("yak_mc_edge_corners"):])
    conn = np.array([[int(a), int(b)] for a, b in pairs], dtype=np.int32)
    assert conn.shape == (12, 2)
    return tri, conn


def marching_cubes(volume, dims, min_weight=4, scale=1.0, tables=None, nthreads=1):
    """Returns triangles float32 (n, 3, 3) in the reference's serial visiting order (nthreads > 1: the x slices in
    parallel, as the reference's OpenMP loop; same order)."""
    dims = np.ascontiguousarray(dims, dtype=np.int32)
    vol = np.ascontiguousarray(volume, dtype=np.uint32).reshape(-1)
    assert vol.size == int(dims[0]) * int(dims[1]) * int(dims[2])
    tri, conn = (tables if tables is not None else mc_tables()[:2])
    tri = np.ascontiguousarray(tri, dtype=np.int32)
    conn = np.ascontiguousarray(conn, dtype=np.int32)
    if nthreads > 1:
        n = lib().yko_marching_cubes_mt(vol, dims, int(min_weight), float(scale), tri.ctypes.data, conn.ctypes.data, None,
                                        0, int(nthreads))
        out = np.empty((n, 3, 3), dtype=np.float32)
        if n:
            assert lib().yko_marching_cubes_mt(vol, dims, int(min_weight), float(scale), tri.ctypes.data,
                                               conn.ctypes.data, out.ctypes.data, n, int(nthreads)) == n
        return out
    n = lib().yko_marching_cubes(vol, dims, int(min_weight), float(scale), tri.ctypes.data, conn.ctypes.data, None, 0)
    out = np.empty((n, 3, 3), dtype=np.float32)
    if n:
        m = lib().yko_marching_cubes(vol, dims, int(min_weight), float(scale), tri.ctypes.data, conn.ctypes.data,
                                     out.ctypes.data, n)
        assert m == n
    return out


def extract_cloud(volume, dims, voxel_size, R, t):
    """Zero crossings between neighbouring voxel centres, mapped by (R, t).  Returns float32 (n, 4)."""
    dims = np.ascontiguousarray(dims, dtype=np.int32)
    vol = np.ascontiguousarray(volume, dtype=np.uint32).reshape(-1)
    assert vol.size == int(dims[0]) * int(dims[1]) * int(dims[2])
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32).reshape(-1)  # noqa: E731
    L = lib()
    L.yko_extract_cloud.restype = C.c_uint64
    L.yko_extract_cloud.argtypes = [_u32p, _i32p, _f32p, _f32p, _f32p, C.c_void_p, C.c_uint64]
    n = L.yko_extract_cloud(vol, dims, f32(voxel_size), f32(R), f32(t), None, 0)
    out = np.empty((n, 4), dtype=np.float32)
    if n:
        assert L.yko_extract_cloud(vol, dims, f32(voxel_size), f32(R), f32(t), out.ctypes.data, n) == n
    return out


def extract_normals(volume, dims, voxel_size, points, R, t, Rinv, gradient_delta_factor=0.5):
    """Normals of extract_cloud's points.  Returns float32 (n, 4), NaN near the volume's faces."""
    dims = np.ascontiguousarray(dims, dtype=np.int32)
    vol = np.ascontiguousarray(volume, dtype=np.uint32).reshape(-1)
    pts = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 4)
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32).reshape(-1)  # noqa: E731
    L = lib()
    L.yko_extract_normals.restype = None
    L.yko_extract_normals.argtypes = [_u32p, _i32p, _f32p, C.c_void_p, C.c_uint64, _f32p, _f32p, _f32p, C.c_float, C.c_void_p]
    out = np.empty_like(pts)
    L.yko_extract_normals(vol, dims, f32(voxel_size), pts.ctypes.data, len(pts), f32(R), f32(t), f32(Rinv),
                          float(gradient_delta_factor), out.ctypes.data)
    return out


def render_image(src, normals, light_pose, intr=None):
    """Phong view of camera-frame maps.  src: points (rows, cols, 4) float32, or depth (rows, cols) uint16
    with intr = the Reprojector's {fx, fy, cx, cy}.  Returns (rows, cols, 4) uint8 {b, g, r, 0}."""
    n = _f4img(normals)
    rows, cols = n.shape[:2]
    img = np.empty((rows, cols, 4), dtype=np.uint8)
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32).reshape(-1)  # noqa: E731
    L = lib()
    vp, sz, i = C.c_void_p, C.c_size_t, C.c_int
    if intr is None:
        p = _f4img(src)
        assert p.shape == n.shape
        L.yko_render_image_points.restype = None
        L.yko_render_image_points.argtypes = [vp, sz, vp, sz, i, i, _f32p, vp, sz]
        L.yko_render_image_points(p.ctypes.data, cols * 16, n.ctypes.data, cols * 16, rows, cols, f32(light_pose),
                                  img.ctypes.data, cols * 4)
    else:
        d = _u16img(src)
        assert d.shape == (rows, cols)
        L.yko_render_image_depth.restype = None
        L.yko_render_image_depth.argtypes = [vp, sz, vp, sz, i, i, _f32p, _f32p, vp, sz]
        L.yko_render_image_depth(d.ctypes.data, cols * 2, n.ctypes.data, cols * 16, rows, cols, f32(intr),
                                 f32(light_pose), img.ctypes.data, cols * 4)
    return img


def render_tangent_colors(normals):
    n = _f4img(normals)
    rows, cols = n.shape[:2]
    img = np.empty((rows, cols, 4), dtype=np.uint8)
    L = lib()
    L.yko_render_tangent_colors.restype = None
    L.yko_render_tangent_colors.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    L.yko_render_tangent_colors(n.ctypes.data, cols * 16, rows, cols, img.ctypes.data, cols * 4)
    return img


def max_threads():
    return int(lib().yko_max_threads())
