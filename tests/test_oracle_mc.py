"""CPU tests: the marching-cubes case table and the oracle's restatement of marchingCubesCPU's core.

PINNED by the reference: the dipyramid known answer (yak/test/marching_cubes_tests.cpp:86-150: a 3^3
volume of +1 with one -1 voxel in the middle meshes to 8 faces over 6 distinct vertices, listed in the
test's PLY text), and -- when oracle/_ref is built -- the reference's own compiled case tables.
"""
import zlib

import numpy as np
import pytest

import oracle

# the vertex list of DIPYRAMID_MESH_ASCII, test/marching_cubes_tests.cpp:24-29 (data, scale 0.01)
DIPYRAMID_VERTICES = np.float32([[0.01, 0.005, 0.01], [0.005, 0.01, 0.01], [0.01, 0.01, 0.005], [0.015, 0.01, 0.01],
                                 [0.01, 0.01, 0.015], [0.01, 0.015, 0.01]])
TABLE_CRC = 0x67D04515  # of yak_amd/csrc/mc_case_table.h, printed by tests/golden/gen_mc_case_table.py


def clean(tris):
    """What MarchingCubesParameters::clean asks of vtkCleanPolyData (marching_cubes.cpp:204-216):
    merge identical points, drop triangles that collapse.  Returns (vertices, faces)."""
    verts, inv = np.unique(tris.reshape(-1, 3), axis=0, return_inverse=True)
    faces = inv.reshape(-1, 3)
    keep = (faces[:, 0] != faces[:, 1]) & (faces[:, 1] != faces[:, 2]) & (faces[:, 0] != faces[:, 2])
    return verts, faces[keep]


def sphere_volume(n, radius_vox, trunc_vox=4.0, weight=4):
    """Config C1: values clamp(d / trunc, -1, 1) as RNE half, constant weight."""
    c = (n - 1) / 2.0
    z, y, x = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    d = np.sqrt((x - c) ** 2 + (y - c) ** 2 + (z - c) ** 2) - radius_vox
    f = np.clip(d / trunc_vox, -1, 1).astype(np.float32)
    return (oracle.float2half(f).astype(np.uint32) | (np.uint32(weight) << 16)).reshape(-1)


def test_packed_case_table_is_the_reference_table():
    tri, conn = oracle.unpack_case_table()
    words = []
    for row in tri:
        n = int((row >= 0).sum())
        w = (n // 3) << 60
        for k in range(n):
            w |= int(row[k]) << (4 * k)
        words.append(w)
    assert zlib.crc32(b"".join(w.to_bytes(8, "little") for w in words)) == TABLE_CRC
    # structural: every case uses only cut edges, and the complementary case has the same edge set
    for case in range(256):
        cut = {e for e in range(12) if ((case >> conn[e, 0]) & 1) != ((case >> conn[e, 1]) & 1)}
        used = set(tri[case][tri[case] >= 0].tolist())
        assert used == cut or (not cut and not used), case
        assert (tri[case] >= 0).sum() % 3 == 0
    R = oracle.ref()
    if R is None:
        pytest.skip("oracle/_ref not built: CRC and structure checked only")
    rtri, rconn, src = oracle.mc_tables()
    assert src == "reference" and np.array_equal(rtri, tri) and np.array_equal(rconn, conn)
    flags = np.ctypeslib.as_array(R.yakref_edge_flags(), shape=(256,))
    for case in range(256):
        derived = sum(1 << e for e in range(12) if ((case >> conn[e, 0]) & 1) != ((case >> conn[e, 1]) & 1))
        assert derived == flags[case]


def test_dipyramid_reference_known_answer():
    # test/marching_cubes_tests.cpp:93-107: 3 x 3 x 3, half(1.0) weight 1 everywhere, half(-1.0) at (1,1,1)
    vol = np.full(27, 0x00013C00, dtype=np.uint32)
    vol[1 + 3 * 1 + 9 * 1] = 0x0001BC00
    tris = oracle.marching_cubes(vol, (3, 3, 3), min_weight=1, scale=0.01)
    assert tris.shape == (8, 3, 3)                       # :121 8 faces
    verts, faces = clean(tris)
    assert len(faces) == 8 and len(verts) == 6            # :133 no duplicate vertices after cleaning
    want = DIPYRAMID_VERTICES[np.lexsort(DIPYRAMID_VERTICES.T[::-1])]
    assert np.array_equal(verts, want)                    # :136-149 same positions (bit for bit here)
    # the packed product table gives the same mesh
    t2 = oracle.marching_cubes(vol, (3, 3, 3), 1, 0.01, tables=oracle.unpack_case_table())
    assert np.array_equal(t2, tris)
    # closed, consistently wound surface: every edge is used once in each direction
    edges = {}
    for f in faces:
        for a, b in ((f[0], f[1]), (f[1], f[2]), (f[2], f[0])):
            edges[(a, b)] = edges.get((a, b), 0) + 1
    assert all(v == 1 for v in edges.values()) and all((b, a) in edges for a, b in edges)


def test_min_weight_masks_cubes():
    vol = np.full(27, 0x00013C00, dtype=np.uint32)
    vol[13] = 0x0001BC00
    assert len(oracle.marching_cubes(vol, (3, 3, 3), min_weight=2)) == 0       # all weights 1 < 2
    vol[0] = 0x00003C00                                                          # corner voxel weight 0
    t = oracle.marching_cubes(vol, (3, 3, 3), min_weight=1)
    assert len(t) == 7                                                           # the cube at (0,0,0) drops out
    # weights are unsigned 16 bit here (tsdf_container.h:45-49): 0xffff is a large weight, not -1
    vol[:] = 0xFFFF3C00
    vol[13] = 0xFFFFBC00
    assert len(oracle.marching_cubes(vol, (3, 3, 3), min_weight=1000)) == 8


def test_degenerate_offsets():
    # equal values across a cut edge cannot happen (cut means different signs); -0.0 is not < 0
    vol = np.full(8, 0x00013C00, dtype=np.uint32)
    vol[0] = 0x00018000                        # half(-0.0): not inside
    assert len(oracle.marching_cubes(vol, (2, 2, 2), 1)) == 0
    vol[0] = 0x0001BC00                        # -1 against +1: vertex exactly half way
    t = oracle.marching_cubes(vol, (2, 2, 2), 1)
    assert t.shape == (1, 3, 3) and sorted(map(tuple, t[0].tolist())) == [(0, 0, 0.5), (0, 0.5, 0), (0.5, 0, 0)]
    assert len(oracle.marching_cubes(vol[:4], (2, 2, 1), 1)) == 0   # no cube fits in one layer


def test_sphere_c1_geometry():
    n, r = 64, 20.3
    vol = sphere_volume(n, r)
    tris = oracle.marching_cubes(vol, (n, n, n), min_weight=4, scale=1.0)
    assert len(tris) > 5000
    c = (n - 1) / 2.0
    dist = np.linalg.norm(tris.reshape(-1, 3) - c, axis=1)
    assert np.abs(dist - r).max() < 0.1                   # linear interpolation of a clamped distance field
    # closed surface: V - E + F = 2 with E = 3F / 2.  Neighbouring cubes walk a shared edge in opposite
    # directions, so the same crossing can come out an ulp apart: merge on a 1e-3 voxel grid for this
    verts, faces = clean(np.round(tris.astype(np.float64) * 1000).astype(np.int64))
    assert len(verts) - 3 * len(faces) // 2 + len(faces) == 2
    # scale is applied in double and narrowed (marching_cubes.cpp:175-183)
    t2 = oracle.marching_cubes(vol, (n, n, n), 4, 0.01)
    assert np.array_equal(t2, (np.float64(0.01) * tris.astype(np.float64)).astype(np.float32))
    assert len(oracle.marching_cubes(vol, (n, n, n), min_weight=5)) == 0


def test_threaded_routine_equals_the_serial_one():
    """yko_marching_cubes_mt (x slices in parallel, the reference's OpenMP loop) gives the serial routine's
    triangles in the serial routine's order; it is what bench.py times at 1 and max threads (config C1)."""
    n = 48
    vol = sphere_volume(n, 15.7)
    a = oracle.marching_cubes(vol, (n, n, n), 4, 0.01)
    b = oracle.marching_cubes(vol, (n, n, n), 4, 0.01, nthreads=4)
    assert len(a) > 1000 and np.array_equal(a, b)
    assert len(oracle.marching_cubes(vol, (n, n, n), 5, 1.0, nthreads=3)) == 0
