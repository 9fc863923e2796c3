"""CPU tests of the cloud-extraction restatement (oracle/yak_oracle.c: yko_extract_cloud / yko_extract_normals;
reference yak/src/cuda/tsdf_volume.cu:464-800).  The reference holds no vector for it (PARITY UNPINNED); pinned
here by analytic known answers derived from the cited lines."""
import numpy as np

import oracle
from test_oracle_mc import sphere_volume


def test_cloud_of_a_planar_field_is_the_plane_between_voxel_centres():
    # tsdf(x, y, z) = clamp((k0 - z + 0.25) / 4, -1, 1) of half-exact values; zero crossing between the centres of
    # layers k0 and k0 + 1 at fraction 0.25 / 1.0 of the way: z = (k0 + 0.5 + 0.25) * vs
    n, k0, vs = 12, 5, 0.1
    z = np.arange(n, dtype=np.float32)
    f = np.clip((k0 - z + 0.25) / 4, -1, 1).astype(np.float16)
    words = f.view(np.uint16).astype(np.uint32) | (3 << 16)
    vol = np.ascontiguousarray(np.broadcast_to(words[:, None, None], (n, n, n))).reshape(-1)
    R = np.eye(3, dtype=np.float32)
    pts = oracle.extract_cloud(vol, (n, n, n), (vs,) * 3, R, np.zeros(3, np.float32))
    assert pts.shape == (n * n, 4)                                    # one dz crossing per column, nothing else
    assert np.abs(pts[:, 2] - (k0 + 0.75) * vs).max() < 1e-6 and (pts[:, 3] == 0).all()
    xy = np.sort(np.round(pts[:, :2] / vs - 0.5).astype(int).view([("x", int), ("y", int)]).reshape(-1))
    assert len(np.unique(xy)) == n * n                                # every column once, at its centre
    # pose: rotate 90 degrees about x and shift: p' = R p + t
    R2 = np.float32([[1, 0, 0], [0, 0, -1], [0, 1, 0]])
    t2 = np.float32([0.5, -1.0, 2.0])
    pts2 = oracle.extract_cloud(vol, (n, n, n), (vs,) * 3, R2, t2)
    assert np.abs(pts2[:, :3] - (pts[:, :3] @ R2.T + t2)).max() < 1e-6
    # weight 0 or tsdf == 1 on either side of a pair removes its point (tsdf_volume.cu:524, 535)
    v2 = vol.reshape(n, n, n).copy()
    v2[k0, 3, 4] &= 0xFFFF                                            # weight 0 below the crossing
    v2[k0 + 1, 7, 8] = 0x00033C00                                     # tsdf 1.0 above it: no crossing left there
    assert len(oracle.extract_cloud(v2.reshape(-1), (n, n, n), (vs,) * 3, R, np.zeros(3, np.float32))) == n * n - 2
    # normals: the field falls along +z only; inside the 2-voxel margin they are defined, outside NaN
    nrm = oracle.extract_normals(vol, (n, n, n), (vs,) * 3, pts, R, np.zeros(3, np.float32), R)
    g = np.round(pts[:, :3] / vs).astype(int)
    inner = ((g > 1) & (g < n - 2)).all(axis=1)
    assert inner.sum() > 30 and np.isnan(nrm[~inner, :3]).all() and (nrm[:, 3] == 0).all()
    assert np.abs(nrm[inner, :3] - np.float32([0, 0, -1])).max() < 1e-6


def test_cloud_of_a_sphere_lies_on_it_with_radial_normals():
    n, r = 64, 20.3
    vol = sphere_volume(n, r)
    R = np.eye(3, dtype=np.float32)
    pts = oracle.extract_cloud(vol, (n, n, n), (1.0, 1.0, 1.0), R, np.zeros(3, np.float32))
    assert len(pts) > 5000
    c = pts[:, :3] - (n / 2)                                          # voxel centres: index + 0.5, sphere centre n/2 - 0.5
    rad = np.linalg.norm(c, axis=1)
    assert np.abs(rad - r).max() < 0.02                               # linear interpolation of a clamped distance field
    nrm = oracle.extract_normals(vol, (n, n, n), (1.0, 1.0, 1.0), pts, R, np.zeros(3, np.float32), R)
    assert not np.isnan(nrm).any()
    assert ((nrm[:, :3] * (c / rad[:, None])).sum(axis=1)).min() > 0.995
