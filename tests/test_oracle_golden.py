"""CPU tests: pin the oracle against every vector the reference holds for this path, and against
hand-computed cases where it holds none.

What the reference pins:
  * voxel wire format  -- yak/test/marching_cubes_tests.cpp:46-79 (0.375 -> 0x3600, weight 19)
  * half -> float      -- the vendored half.hpp, compiled from /root/reference into oracle/_ref
Everything else about integrate / compute_dists is PARITY UNPINNED in the reference (it has no
test for them); the cases below are hand-computed from the cited CUDA source lines.
"""
import zlib

import numpy as np
import pytest

import oracle
from yak_amd import synth


# ---------------------------------------------------------------------------- wire format / half
def test_pack_unpack_reference_known_answer():
    # test/marching_cubes_tests.cpp:63-79: half(0.375f), weight 19
    L = oracle.lib()
    word = L.yko_pack_tsdf(0.375, 19)
    assert word == 0x00133600
    import ctypes

    w = ctypes.c_int(0)
    assert L.yko_unpack_tsdf(word, ctypes.byref(w)) == 0.375
    assert w.value == 19
    # clear value, tsdf_volume.cu:25
    assert L.yko_pack_tsdf(0.0, 0) == 0


def test_wire_format_matches_reference_pack_unpack():
    R = oracle.ref()
    if R is None:
        pytest.skip("oracle/_ref not built (reference tree absent)")
    import ctypes

    L = oracle.lib()
    # values exactly representable in half: half.hpp's truncating ctor and RNE agree on them
    for dist, weight in [(0.375, 19), (1.0, 1), (-1.0, 1), (0.0, 0), (-0.5, 64), (0.999023438, 32767)]:
        assert R.yakref_pack(dist, weight) == L.yko_pack_tsdf(dist, weight)
        d, w = ctypes.c_float(), ctypes.c_uint16()
        R.yakref_unpack(L.yko_pack_tsdf(dist, weight), ctypes.byref(d), ctypes.byref(w))
        wi = ctypes.c_int()
        assert L.yko_unpack_tsdf(L.yko_pack_tsdf(dist, weight), ctypes.byref(wi)) == d.value
        assert wi.value == w.value


def test_half2float_all_codes_vs_reference_half_hpp():
    R = oracle.ref()
    if R is None:
        pytest.skip("oracle/_ref not built (reference tree absent)")
    bits = np.arange(65536, dtype=np.uint16)
    mine = oracle.half2float(bits)
    theirs = np.array([R.yakref_half2float(int(b)) for b in bits], dtype=np.float32)
    nan = np.isnan(theirs)
    assert np.array_equal(np.isnan(mine), nan)
    assert np.array_equal(mine[~nan].view(np.uint32), theirs[~nan].view(np.uint32))


def test_reference_half_ctor_truncates_so_oracle_does_not_use_it():
    # SURVEY.md probe: half.hpp:214-215 rounds toward zero; __float2half is RNE.
    R = oracle.ref()
    if R is None:
        pytest.skip("oracle/_ref not built")
    f = np.float32(0.1)  # 0x2E66 (RNE) vs 0x2E66/0x2E65 depending on mode
    x = np.float32(1.0 + 3 * 2.0 ** -12)  # 0.75 ulp above 1.0: RNE rounds up, truncation does not
    assert int(oracle.float2half(x)) == 0x3C01
    assert R.yakref_float2half_ctor(float(x)) == 0x3C00
    assert int(oracle.float2half(f)) == int(np.float16(f).view(np.uint16))


def test_float2half_rne_vs_numpy_and_edges():
    rng = np.random.default_rng(7)
    f = rng.integers(0, 2 ** 32, size=2_000_000, dtype=np.uint64).astype(np.uint32).view(np.float32)
    f = np.concatenate([f, rng.uniform(-1.5, 1.5, 500_000).astype(np.float32),
                        # ties and near-ties around every half in [0.5, 2)
                        (np.arange(0x3800, 0x4000, dtype=np.uint16).view(np.float16).astype(np.float32)
                         + np.float32(2.0 ** -12)),
                        np.array([0.0, -0.0, 65504.0, 65519.99, 65520.0, 1e9, 2.0 ** -24, 2.0 ** -25,
                                  np.nextafter(np.float32(2.0 ** -25), np.float32(1)), 2.0 ** -14, 6.0e-8,
                                  np.inf, -np.inf], dtype=np.float32)])
    mine = oracle.float2half(f)
    with np.errstate(over="ignore", invalid="ignore"):
        ref = f.astype(np.float16).view(np.uint16)
    nan = np.isnan(f)
    assert np.array_equal(mine[~nan], ref[~nan])
    assert np.all(np.isnan(mine[nan].view(np.float16)))
    # round trip of every finite half is the identity
    bits = np.arange(65536, dtype=np.uint16)
    fin = ~np.isnan(bits.view(np.float16))
    assert np.array_equal(oracle.float2half(oracle.half2float(bits))[fin], bits[fin])


# ---------------------------------------------------------------------------- compute_dists
def test_compute_dists_hand_cases():
    # imgproc.cu:266-270 at the principal point lambda = 1, so dists = half(depth * 0.001)
    intr = (100.0, 100.0, 2.0, 1.0)
    depth = np.zeros((3, 5), dtype=np.uint16)
    depth[1, 2] = 1000   # centre pixel: exactly 1.0 m
    depth[1, 4] = 2000   # xl = 0.02, yl = 0
    depth[0, 0] = 0      # no return stays 0
    d = oracle.compute_dists(depth, intr)
    assert d[1, 2] == 0x3C00
    xl = (np.float32(4) - np.float32(2)) * (np.float32(1) / np.float32(100))
    lam = np.sqrt(xl * xl + np.float32(0) + np.float32(1), dtype=np.float32)
    expect = np.float16(np.float32(2000) * lam * np.float32(0.001))
    assert d[1, 4] == expect.view(np.uint16)
    assert d[0, 0] == 0 and np.count_nonzero(d) == 2


def test_compute_dists_independent_numpy_restatement():
    depth, _, _ = synth.scene_s1()
    fx, fy, cx, cy = (np.float32(v) for v in synth.INTR)
    x = np.arange(synth.COLS, dtype=np.float32)[None, :]
    y = np.arange(synth.ROWS, dtype=np.float32)[:, None]
    xl = (x - cx) * (np.float32(1) / fx)
    yl = (y - cy) * (np.float32(1) / fy)
    lam = np.sqrt(xl * xl + yl * yl + np.float32(1)).astype(np.float32)
    expect = ((depth.astype(np.float32) * lam) * np.float32(0.001)).astype(np.float16).view(np.uint16)
    assert np.array_equal(oracle.compute_dists(depth, synth.INTR), expect)


# ---------------------------------------------------------------------------- integrate
def _column_case(dists_value_m, n=10, trunc=0.25, max_weight=64, frames=1, start=None):
    """A 1x1xN column on the optical axis: R = I, t = (0,0,0.5), voxel 0.1 m, 4x4 image with
    fx = fy = 100, c = 1.5, so every voxel projects to pixel (1,1) and sdf = D - z_k."""
    dims = (1, 1, n)
    vs = (0.1, 0.1, 0.1)
    dists = np.zeros((4, 4), dtype=np.uint16)
    dists[1, 1] = oracle.float2half(np.float32(dists_value_m))
    vol = np.zeros(n, dtype=np.uint32) if start is None else start.copy()
    R = np.eye(3, dtype=np.float32)
    t = np.array([0, 0, 0.5], dtype=np.float32)
    stats = (0, 0)
    for _ in range(frames):
        stats = oracle.integrate(vol, dims, vs, trunc, max_weight, dists, R, t, (100.0, 100.0, 1.5, 1.5))
    return vol, stats


def test_integrate_hand_computed_column():
    vol, (visited, updated) = _column_case(1.0)
    # z_k accumulated in float32 exactly as tsdf_volume.cu:76 does
    z = np.float32(0.5)
    step = np.float32(1.0) * np.float32(0.1)
    expect = []
    for _ in range(10):
        sdf = np.float32(1.0) - np.sqrt(np.float32(z * z), dtype=np.float32)
        if sdf >= -np.float32(0.25):
            tsdf = min(np.float32(1.0), np.float32(sdf * (np.float32(1) / np.float32(0.25))))
            expect.append((int(np.float16(tsdf).view(np.uint16)), 1))
        else:
            expect.append((0, 0))
        z = np.float32(z + step)
    got = [(int(v & 0xFFFF), int(v >> 16)) for v in vol]
    assert got == expect
    assert visited == 10 and updated == sum(1 for e in expect if e[1])
    # the five regimes named in SURVEY 8c(i)
    assert got[0] == (0x3C00, 1)            # in front by more than trunc: tsdf 1.0, w 1
    assert got[5][1] == 1 and abs(np.uint16(got[5][0]).view(np.float16)) < 1e-3  # on the surface: ~0
    assert np.uint16(got[7][0]).view(np.float16) < 0  # just behind, inside the band: negative
    assert got[9] == (0, 0)                 # behind by more than trunc: untouched


def test_integrate_no_measurement_and_out_of_image():
    # Dp == 0 (tsdf_volume.cu:89): nothing changes
    vol, (_, updated) = _column_case(0.0)
    assert updated == 0 and not vol.any()
    # camera looking away (vc.z <= 0 for every voxel): nothing changes
    dims, vs = (1, 1, 4), (0.1, 0.1, 0.1)
    dists = np.full((4, 4), 0x3C00, dtype=np.uint16)
    vol = np.zeros(4, dtype=np.uint32)
    _, updated = oracle.integrate(vol, dims, vs, 0.25, 64, dists, np.eye(3, dtype=np.float32),
                                  np.array([0, 0, -1.0], dtype=np.float32), (100.0, 100.0, 1.5, 1.5))
    assert updated == 0 and not vol.any()
    # projection outside the 4x4 image (:83): t.x shifts the column 1 m sideways
    _, updated = oracle.integrate(vol, dims, vs, 0.25, 64, dists, np.eye(3, dtype=np.float32),
                                  np.array([1.0, 0, 0.5], dtype=np.float32), (100.0, 100.0, 1.5, 1.5))
    assert updated == 0 and not vol.any()


def test_integrate_weight_saturates_at_max_weight():
    vol, _ = _column_case(1.0, n=2, max_weight=3, frames=7)
    assert [int(v >> 16) for v in vol] == [3, 3]     # min(Wp + 1, max_weight), tsdf_volume.cu:103
    assert [int(v & 0xFFFF) for v in vol] == [0x3C00, 0x3C00]  # mean of ones stays 1.0


def test_integrate_running_average_two_frames():
    # frame 1 with D = 1.0 then frame 2 with D = 0.9: voxel k=4 (z ~ 0.9) averages tsdf 0.4 and 0.0
    vol, _ = _column_case(1.0)
    dists2 = np.float32(0.9)
    vol2, _ = _column_case(dists2, start=vol)
    z = np.float32(0.5)
    for _ in range(4):
        z = np.float32(z + np.float32(0.1))
    inv = np.float32(1) / np.float32(0.25)
    t1 = np.float16(min(np.float32(1), np.float32((np.float32(1.0) - z) * inv)))
    d2 = np.float16(dists2).astype(np.float32)
    t2 = min(np.float32(1), np.float32((d2 - z) * inv))
    fn = np.float32(np.float32(np.float32(t1) * np.float32(1) + t2) / np.float32(2))  # fma exact here? check below
    # the fma(Fp, 1, t2) == Fp + t2 rounded once, which is what the line above computes
    assert int(vol2[4] & 0xFFFF) == int(np.float16(fn).view(np.uint16))
    assert int(vol2[4] >> 16) == 2


def test_integrate_slab_equals_whole_volume_layers():
    depth, R, t = synth.scene_s2(17)
    dists = oracle.compute_dists(depth, synth.INTR)
    dims, vs, tr = synth.volume_geometry((24, 20, 28))
    whole = np.zeros(24 * 20 * 28, dtype=np.uint32)
    oracle.integrate(whole, dims, vs, tr, 64, dists, R, t, synth.INTR)
    assert whole.any()
    for z0, z1 in [(0, 9), (9, 21), (21, 28)]:
        slab = np.zeros(24 * 20 * (z1 - z0), dtype=np.uint32)
        oracle.integrate(slab, dims, vs, tr, 64, dists, R, t, synth.INTR, z0=z0, z1=z1)
        assert np.array_equal(slab, whole[z0 * 480:z1 * 480])


def test_integrate_threads_agree_with_scalar():
    depth, R, t = synth.scene_s1()
    dists = oracle.compute_dists(depth, synth.INTR)
    dims, vs, tr = synth.volume_geometry(48)
    a = np.zeros(48 ** 3, dtype=np.uint32)
    b = np.zeros(48 ** 3, dtype=np.uint32)
    sa = oracle.integrate(a, dims, vs, tr, 64, dists, R, t, synth.INTR, nthreads=1)
    sb = oracle.integrate(b, dims, vs, tr, 64, dists, R, t, synth.INTR, nthreads=4)
    assert sa == sb and np.array_equal(a, b)


def test_scene_s0_updates_every_voxel_with_tsdf_one():
    # the bench workload's defining property: N_updated = N^3, tsdf = 1, w = 1
    depth, R, t = synth.scene_s0()
    dists = oracle.compute_dists(depth, synth.INTR)
    dims, vs, tr = synth.volume_geometry(64)
    vol = np.zeros(64 ** 3, dtype=np.uint32)
    visited, updated = oracle.integrate(vol, dims, vs, tr, 64, dists, R, t, synth.INTR, nthreads=4)
    assert visited == updated == 64 ** 3
    assert np.all(vol == 0x00013C00)


# ---------------------------------------------------------------------------- committed fixtures
def test_golden_fixture_s1_32():
    """tests/golden/s1_32.npz was written by tests/golden/make_golden.py from this oracle (the
    reference cannot produce it: CUDA-only).  It guards the oracle itself against drift."""
    import os

    path = os.path.join(os.path.dirname(__file__), "golden", "s1_32.npz")
    g = np.load(path)
    depth, R, t = synth.scene_s1()
    assert zlib.crc32(depth.tobytes()) == int(g["depth_crc"])
    dists = oracle.compute_dists(depth, synth.INTR)
    assert zlib.crc32(dists.tobytes()) == int(g["dists_crc"])
    dims, vs, tr = synth.volume_geometry(32)
    vol = np.zeros(32 ** 3, dtype=np.uint32)
    oracle.integrate(vol, dims, vs, tr, 64, dists, R, t, synth.INTR)
    assert np.array_equal(vol, g["vol_f1"])
    for _ in range(2):
        oracle.integrate(vol, dims, vs, tr, 64, dists, R, t, synth.INTR)
    assert np.array_equal(vol, g["vol_f3"])


def test_oracle_rejects_nan_pixel_coordinates_instead_of_indexing_with_them():
    """The stated deviation in integrate_column (oracle/yak_oracle.c): a NaN pixel coordinate -- non-finite pose or
    intrinsics -- skips the voxel; the reference's __float2int_rd would turn it into 0 and read row / column 0.
    (Before this was written out, the C restatement indexed the image with INT_MIN.)"""
    import oracle
    from yak_amd import synth

    depth, R, t = synth.scene_s2(40)
    dists = oracle.compute_dists(depth, synth.INTR)
    dims, vs, tr = synth.volume_geometry((24, 20, 28))
    R = np.asarray(R, dtype=np.float32).reshape(3, 3)
    t = np.asarray(t, dtype=np.float32)
    for R2, t2, intr in [(R, np.array([t[0], np.nan, t[2]], dtype=np.float32), synth.INTR),
                         (np.where(np.eye(3) > 0, R, np.float32(np.nan)).astype(np.float32), t, synth.INTR),
                         (R, t, (synth.INTR[0], synth.INTR[1], float("nan"), synth.INTR[3])),
                         (R, t, (float("nan"), float("nan"), float("nan"), float("nan")))]:
        vol = np.zeros(24 * 20 * 28, dtype=np.uint32)
        _, updated = oracle.integrate(vol, dims, vs, tr, synth.MAX_WEIGHT, dists, R2, t2, intr)
        assert updated == 0 and not vol.any()
    vol = np.zeros(24 * 20 * 28, dtype=np.uint32)
    _, updated = oracle.integrate(vol, dims, vs, tr, synth.MAX_WEIGHT, dists, R, t, synth.INTR)
    assert updated > 0
