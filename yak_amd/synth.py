"""Deterministic synthetic depth frames for benchmarks and parity tests.

Analytic scenes (planes + spheres) ray-cast in float64 and quantised the way a depth sensor
delivers them to `FusionServer::fuse`: uint16 millimetres along the optical axis, 0 = no return.
No RNG unless `noise_mm` is given.
"""
import math

import numpy as np

# KinFuParams::default_params (reference yak/src/kfusion/kinfu.cpp:17-40)
ROWS, COLS = 480, 640
INTR = (580.0, 580.0, COLS / 2 - 0.5, ROWS / 2 - 0.5)
VOLUME_SIZE_M = 3.0
TRUNC_DIST = 0.04
MAX_WEIGHT = 64


def volume_geometry(n, size_m=VOLUME_SIZE_M, trunc=TRUNC_DIST):
    """(dims, voxel_size, trunc_dist) as the reference host code derives them in float32:
    voxel = size / dims (tsdf_volume.cpp:45-48); trunc = max(trunc, 2.1 * max voxel) (:62-67)."""
    dims = (n, n, n) if np.isscalar(n) else tuple(int(v) for v in n)
    size = np.float32(size_m)
    vs = tuple(np.float32(size / np.float32(d)) for d in dims)
    tr = max(np.float32(trunc), np.float32(np.float32(2.1) * max(vs)))
    return dims, tuple(float(v) for v in vs), float(tr)


def look_at(eye, target, up=(0.0, -1.0, 0.0)):
    """Camera-to-volume pose (R_c2v 3x3, t = eye) with +z forward, +x right, +y down (image axes)."""
    eye = np.asarray(eye, dtype=np.float64)
    fwd = np.asarray(target, dtype=np.float64) - eye
    fwd /= np.linalg.norm(fwd)
    down = -np.asarray(up, dtype=np.float64)
    right = np.cross(down, fwd)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    R = np.stack([right, down, fwd], axis=1)
    return R, eye


def vol2cam(R_c2v, eye):
    """The (R, t) `TsdfVolume::integrate` hands to the device (tsdf_volume.cpp:97-104 with pose_ = I):
    the inverse of the camera pose, as float32 row-major 3x3 + translation."""
    R = np.asarray(R_c2v, dtype=np.float64).T
    t = -R @ np.asarray(eye, dtype=np.float64)
    return R.astype(np.float32), t.astype(np.float32)


def render_depth(R_c2v, eye, planes=(), spheres=(), intr=INTR, rows=ROWS, cols=COLS, noise_mm=0.0, seed=12345):
    """planes: (point, normal) pairs; spheres: (centre, radius) pairs -- all in volume coordinates.
    Returns (rows, cols) uint16 depth in mm (camera-space z of the nearest hit), 0 where no hit."""
    fx, fy, cx, cy = intr
    u, v = np.meshgrid(np.arange(cols, dtype=np.float64), np.arange(rows, dtype=np.float64))
    d_cam = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], axis=-1)  # z component 1: s == depth
    R = np.asarray(R_c2v, dtype=np.float64)
    d = d_cam @ R.T
    o = np.asarray(eye, dtype=np.float64)
    best = np.full((rows, cols), np.inf)

    for point, normal in planes:
        n = np.asarray(normal, dtype=np.float64)
        denom = d @ n
        num = (np.asarray(point, dtype=np.float64) - o) @ n
        with np.errstate(divide="ignore", invalid="ignore"):
            s = num / denom
        s = np.where((np.abs(denom) > 1e-12) & (s > 1e-6), s, np.inf)
        best = np.minimum(best, s)

    for centre, radius in spheres:
        oc = o - np.asarray(centre, dtype=np.float64)
        a = np.einsum("...i,...i", d, d)
        b = 2.0 * (d @ oc)
        c = oc @ oc - radius * radius
        disc = b * b - 4 * a * c
        with np.errstate(invalid="ignore"):
            root = np.sqrt(np.where(disc >= 0, disc, np.nan))
            s = (-b - root) / (2 * a)
        s = np.where((disc >= 0) & (s > 1e-6), s, np.inf)
        best = np.minimum(best, s)

    mm = best * 1000.0
    if noise_mm > 0:
        rng = np.random.default_rng(seed)
        mm = mm + rng.normal(0.0, noise_mm, size=mm.shape)
    mm = np.where(np.isfinite(mm) & (mm < 65535.0) & (mm > 0), np.rint(mm), 0.0)
    return mm.astype(np.uint16)


# ---- the named scenes of SURVEY.md section 8d -------------------------------------------------

def scene_s0():
    """'wall': camera 3.7 m in front of the cube looking down +z, a wall at z_cam = 7 m.
    Every voxel of the 3 m cube projects inside the image and lies in front of the wall by more than
    the truncation distance, so every voxel updates (N_updated = N^3): the pure-bandwidth case."""
    R_c2v, eye = np.eye(3), np.array([1.5, 1.5, -3.7])
    depth = render_depth(R_c2v, eye, planes=[((0, 0, 3.3), (0, 0, 1))])
    R, t = vol2cam(R_c2v, eye)
    return depth, R, t


def scene_s1():
    """'sphere + wall': S0 plus a 1 m sphere at the cube centre (surface crossings, occlusion)."""
    R_c2v, eye = np.eye(3), np.array([1.5, 1.5, -3.7])
    depth = render_depth(R_c2v, eye, planes=[((0, 0, 3.3), (0, 0, 1))], spheres=[((1.5, 1.5, 1.5), 1.0)])
    R, t = vol2cam(R_c2v, eye)
    return depth, R, t


S2_SPHERES = [((1.1, 1.45, 1.2), 0.5), ((2.0, 1.6, 1.4), 0.35), ((1.5, 1.7, 2.1), 0.25)]
S2_GROUND = ((0.0, 1.95, 0.0), (0.0, 1.0, 0.0))  # plane y = 1.95 (y is 'down' in the camera convention)


def scene_s2(frame, n_frames=300, radius=3.2, noise_mm=0.0):
    """'orbit': frame `frame` of a 1.2 deg/frame circle of radius 3.2 m around the cube centre at
    height y = 1.5, looking at the centre; ground plane + three spheres of different radii."""
    ang = math.radians(1.2 * frame)
    centre = np.array([1.5, 1.5, 1.5])
    eye = centre + np.array([radius * math.sin(ang), 0.0, -radius * math.cos(ang)])
    R_c2v, eye = look_at(eye, centre)
    depth = render_depth(R_c2v, eye, planes=[S2_GROUND], spheres=S2_SPHERES, noise_mm=noise_mm, seed=12345 + frame)
    R, t = vol2cam(R_c2v, eye)
    return depth, R, t


def sphere_tsdf_volume(n, radius_vox, trunc_vox=4.0, weight=4):
    """Config C1 of BASELINE.json: the analytic sphere of the reference's stand-alone marching-cubes program
    (yak/src/mc/marching_cubes_tests.cpp:19-43: radius 25.6 voxels in 256^3) as a packed TSDF volume --
    clamp(d / trunc, -1, 1) rounded to half (nearest even), constant weight.  uint32, x fastest."""
    c = (n - 1) / 2.0
    z, y, x = np.meshgrid(np.arange(n, dtype=np.float32), np.arange(n, dtype=np.float32),
                          np.arange(n, dtype=np.float32), indexing="ij")
    d = np.sqrt((x - c) ** 2 + (y - c) ** 2 + (z - c) ** 2) - np.float32(radius_vox)
    f = np.clip(d / np.float32(trunc_vox), -1, 1).astype(np.float16)
    return (f.view(np.uint16).astype(np.uint32) | (np.uint32(weight) << 16)).reshape(-1)
