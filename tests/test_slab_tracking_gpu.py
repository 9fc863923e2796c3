"""GPU parity of the two tracking-side steps that exchange data when the volume is spread over GPUs
(SURVEY.md 8e): yakhip_raycast_slab + composite, yakhip_icp_accumulate_rows + sum.  One GPU plays every
rank in turn: each "rank" gets its own copy of the layers it would hold (own + halo), nothing else.

Bars: per-slab keys and payloads bit-identical to the oracle's slab march; the composite bit-identical to
the whole-volume ray cast (HIP and oracle); ICP counts exact, sums within 1e-4 relative of the one-launch
sums (the order of the float additions differs)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

import oracle  # noqa: E402
from test_raycast_gpu import cam2vol, fused_volume, gpu_raycast  # noqa: E402
from yak_amd import capi, synth  # noqa: E402
from yak_amd.shard import KEY_NONE, composite_raycast, held_range, row_range, slab_range  # noqa: E402

pytestmark = pytest.mark.gpu

SEEN = {"deferred": 0, "deferred_winners": 0}   # filled by the composite tests, checked by the last one of them


def slab_raycast(ctx, vol_host, dims, vs, tr, z_own, halo, R, t, Rinv, intr, rows, cols, want_depth=False):
    """What one rank does: holds [own - halo, own + halo) only."""
    h0, h1 = held_range(dims[2], z_own[0], z_own[1], halo)
    plane = dims[0] * dims[1]
    held = torch.from_numpy(vol_host[h0 * plane:h1 * plane].view(np.int32).copy()).cuda()
    desc = capi.volume_desc(held.data_ptr(), dims, vs, tr, synth.MAX_WEIGHT, h0, h1)
    keys = torch.full((rows, cols), -5, dtype=torch.int32, device="cuda")
    nrm = torch.full((rows, cols, 4), 7.0, dtype=torch.float32, device="cuda")
    if want_depth:
        img = torch.full((rows, cols), 7, dtype=torch.int16, device="cuda")
        ctx.raycast_slab(desc, z_own[0], z_own[1], R.reshape(-1), t, Rinv.reshape(-1), intr, keys.data_ptr(), cols * 4,
                         nrm.data_ptr(), cols * 16, rows, cols, depth_ptr=img.data_ptr(), depth_pitch=cols * 2)
    else:
        img = torch.full((rows, cols, 4), 7.0, dtype=torch.float32, device="cuda")
        ctx.raycast_slab(desc, z_own[0], z_own[1], R.reshape(-1), t, Rinv.reshape(-1), intr, keys.data_ptr(), cols * 4,
                         nrm.data_ptr(), cols * 16, rows, cols, points_ptr=img.data_ptr(), points_pitch=cols * 16)
    torch.cuda.synchronize()
    return img, nrm, keys, (held, desc, z_own)


def composite_on_one_gpu(ctx, parts, want_depth, Rinv):
    """composite_raycast's two steps without the collectives: the smallest key owns the pixel; then every
    "rank" runs yakhip_raycast_resolve on its copy of the composite and the pixels it changed are taken over."""
    keys = torch.stack([k for _, _, k, _ in parts])
    best, win = keys.min(dim=0)
    assert int(((keys == best).sum(dim=0)[best != KEY_NONE] != 1).sum()) == 0       # no ties
    imgs = torch.stack([i for i, _, _, _ in parts])
    nrms = torch.stack([n for _, n, _, _ in parts])
    if want_depth:
        img = torch.gather(imgs, 0, win[None])[0]
        img = torch.where(best == KEY_NONE, torch.zeros_like(img), img)
    else:
        img = torch.gather(imgs, 0, win[None, ..., None].expand(1, *imgs.shape[1:]))[0]
        img = torch.where((best == KEY_NONE)[..., None], torch.full_like(img, float("nan")), img)
    nrm = torch.gather(nrms, 0, win[None, ..., None].expand(1, *nrms.shape[1:]))[0]
    nrm = torch.where((best == KEY_NONE)[..., None], torch.full_like(nrm, float("nan")), nrm)

    odd = (best != KEY_NONE) & ((best & 1) == 1)
    n_odd = int(odd.sum())
    rows, cols = best.shape
    fin_img, fin_nrm, fin_keys = img.clone(), nrm.clone(), best.clone()
    for _, _, _, (held, desc, own) in parts:
        k_r, i_r, n_r = best.clone(), img.clone().contiguous(), nrm.clone().contiguous()
        kw = dict(depth_ptr=i_r.data_ptr(), depth_pitch=cols * 2) if want_depth else dict(points_ptr=i_r.data_ptr(),
                                                                                          points_pitch=cols * 16)
        ctx.raycast_resolve(desc, own[0], own[1], Rinv.reshape(-1), k_r.data_ptr(), cols * 4, n_r.data_ptr(), cols * 16, rows,
                            cols, **kw)
        torch.cuda.synchronize()
        took = odd & ((k_r & 1) == 0)
        assert int((took & ((fin_keys & 1) == 0)).sum()) == 0                      # one slab per pixel
        assert int((k_r != best)[~took].sum()) == 0 and torch.equal(n_r.view(torch.int32)[~took], nrm.view(torch.int32)[~took])
        fin_keys = torch.where(took, k_r, fin_keys)
        fin_img = torch.where(took if want_depth else took[..., None], i_r, fin_img)
        fin_nrm = torch.where(took[..., None], n_r, fin_nrm)
    assert int((fin_keys[fin_keys != KEY_NONE] & 1).sum()) == 0                    # every deferred normal was computed
    return fin_img, fin_nrm, fin_keys, n_odd


@pytest.mark.parametrize("n,world,view", [(64, 2, 10), (64, 4, 130), (128, 8, 250), (96, 3, 77)])
def test_slab_raycasts_match_the_oracle_and_composite_to_the_whole(gpu_ctx, n, world, view):
    vol, dims, vs, tr = fused_volume(n, [0, 60, 120, 180, 240], scene="s2")
    rows, cols, intr = 120, 160, (145.0, 145.0, 79.5, 59.5)
    _, R_v2c, t_v2c = synth.scene_s2(view)
    R, t = cam2vol(R_v2c, t_v2c)
    Rinv = np.ascontiguousarray(R.T)
    halo = capi.raycast_halo(capi.volume_desc(256, dims, vs, tr, synth.MAX_WEIGHT))
    assert halo == int(np.ceil(tr * 0.75 / vs[2])) + 1 + 2

    parts, deferred = [], 0
    for rank in range(world):
        own = slab_range(dims[2], world, rank)
        img, nrm, keys, slab = slab_raycast(gpu_ctx, vol, dims, vs, tr, own, halo, R, t, Rinv, intr, rows, cols)
        w_img, w_nrm, w_keys = oracle.raycast_slab(vol, dims, vs, tr, R, t, Rinv, intr, rows, cols, own[0], own[1], nthreads=8)
        k = keys.cpu().numpy()
        assert np.array_equal(k & ~1, w_keys & ~1) and ((k == KEY_NONE) == (w_keys == KEY_NONE)).all()   # the same pair
        final = (k == KEY_NONE) | ((k & 1) == 0)                                  # (odd: normal deferred to another slab)
        deferred += int((~final).sum())
        assert np.array_equal(img.cpu().numpy().view(np.uint32)[final], w_img.view(np.uint32)[final])
        assert np.array_equal(nrm.cpu().numpy().view(np.uint32)[final], w_nrm.view(np.uint32)[final])
        parts.append((img, nrm, keys, slab))

    c_img, c_nrm, best, n_odd = composite_on_one_gpu(gpu_ctx, parts, False, Rinv)
    assert n_odd <= deferred <= 0.01 * rows * cols
    SEEN["deferred"] += deferred
    SEEN["deferred_winners"] += n_odd
    whole_p, whole_n = gpu_raycast(gpu_ctx, vol, dims, vs, tr, R, t, Rinv, intr, rows, cols)
    o_p, o_n, _, hits = oracle.raycast(vol, dims, vs, tr, R, t, Rinv, intr, rows, cols, nthreads=8)
    assert hits > 1500
    for got, want in ((c_img, whole_p), (c_nrm, whole_n), (c_img, o_p), (c_nrm, o_n)):
        assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32))
    # shard.composite_raycast with one rank is the identity on a whole-volume "slab"
    (same,), _, unresolved = composite_raycast(None, best, [c_img], [float("nan")])
    assert unresolved == 0 and np.array_equal(same.cpu().numpy().view(np.uint32), whole_p.view(np.uint32))


def test_deferred_normals_were_exercised():
    """The parametrised cases above must have gone through yakhip_raycast_resolve for real pixels."""
    assert SEEN["deferred"] > 0 and SEEN["deferred_winners"] > 0, SEEN


def test_slab_raycast_depth_overload_and_uneven_slabs(gpu_ctx):
    vol, dims, vs, tr = fused_volume((80, 64, 72), [30, 100, 170], scene="s2")
    rows, cols, intr = 97, 131, (120.0, 118.0, 64.2, 47.9)
    _, R_v2c, t_v2c = synth.scene_s2(60)
    R, t = cam2vol(R_v2c, t_v2c)
    Rinv = np.ascontiguousarray(R.T)
    halo = capi.raycast_halo(capi.volume_desc(256, dims, vs, tr, synth.MAX_WEIGHT))
    parts = [slab_raycast(gpu_ctx, vol, dims, vs, tr, own, halo, R, t, Rinv, intr, rows, cols, want_depth=True)
             for own in ((0, 5), (5, 6), (6, 50), (50, 72))]
    c_dep, c_nrm, _, _ = composite_on_one_gpu(gpu_ctx, parts, True, Rinv)
    w_dep, w_nrm = gpu_raycast(gpu_ctx, vol, dims, vs, tr, R, t, Rinv, intr, rows, cols, want_depth=True)
    assert (w_dep > 0).sum() > 1000
    assert np.array_equal(c_dep.cpu().numpy().view(np.uint16), w_dep)
    assert np.array_equal(c_nrm.cpu().numpy().view(np.uint32), w_nrm.view(np.uint32))


def test_slab_raycast_rejects_thin_halos_and_bad_ranges(gpu_ctx):
    dims, vs, tr = synth.volume_geometry(64)
    held = torch.zeros(64 * 64 * 30, dtype=torch.int32, device="cuda")
    keys = torch.zeros((8, 8), dtype=torch.int32, device="cuda")
    nrm = torch.zeros((8, 8, 4), dtype=torch.float32, device="cuda")
    pts = torch.zeros((8, 8, 4), dtype=torch.float32, device="cuda")
    eye, t0 = np.eye(3, dtype=np.float32).reshape(-1), np.zeros(3, dtype=np.float32)

    def call(h0, h1, o0, o1, keys_ptr=None):
        desc = capi.volume_desc(held.data_ptr(), dims, vs, tr, 64, h0, h1)
        gpu_ctx.raycast_slab(desc, o0, o1, eye, t0, eye, synth.INTR, keys.data_ptr() if keys_ptr is None else keys_ptr, 32,
                             nrm.data_ptr(), 128, 8, 8, points_ptr=pts.data_ptr(), points_pitch=128)

    halo = capi.raycast_halo(capi.volume_desc(256, dims, vs, tr, 64))
    call(10, 40, 10 + halo, 40 - halo)                                          # exactly the halo: fine
    call(0, 30, 0, 30 - halo)                                                   # the volume's end needs no halo
    for bad in ((10, 40, 10 + halo - 1, 40 - halo), (10, 40, 10 + halo, 40 - halo + 1), (10, 40, 20, 20), (10, 40, 5, 30),
                (10, 40, 20, 41)):
        with pytest.raises(capi.YakHipError):
            call(*bad)
    with pytest.raises(capi.YakHipError):
        call(10, 40, 10 + halo, 40 - halo, keys_ptr=0)


def tracking_maps(n_frames=(0, 40, 80)):
    """Current maps from a depth frame, model maps from a ray cast of the fused volume: one ICP problem."""
    vol, dims, vs, tr = fused_volume(96, list(n_frames), scene="s2")
    depth, R_v2c, t_v2c = synth.scene_s2(40)
    R, t = cam2vol(R_v2c, t_v2c)
    vprev, nprev, _, _ = oracle.raycast(vol, dims, vs, tr, R, t, np.ascontiguousarray(R.T), synth.INTR, synth.ROWS, synth.COLS,
                                        nthreads=8)
    vcurr, ncurr = oracle.compute_points_normals(oracle.bilateral_filter(depth), synth.INTR)
    return vcurr, ncurr, vprev, nprev


@pytest.mark.parametrize("world", [2, 5, 8])
def test_icp_row_bands_sum_to_the_whole_image(gpu_ctx, world):
    vcurr, ncurr, vprev, nprev = tracking_maps()
    rows, cols = vcurr.shape[:2]
    dev = [torch.from_numpy(a).cuda() for a in (vcurr, ncurr, vprev, nprev)]
    Rg = np.eye(3, dtype=np.float32).reshape(-1)
    tg = np.float32([0.003, -0.002, 0.004])
    args = [a for d in dev for a in (d.data_ptr(), cols * 16)]
    w_sums, w_counts = gpu_ctx.icp_accumulate(Rg, tg, synth.INTR, 0, 0.1, 0.6, *args, rows, cols)
    assert w_counts[0] > 50000 and sum(w_counts) == rows * cols

    total = torch.zeros(33, dtype=torch.float32, device="cuda")
    for rank in range(world):
        r0, r1 = row_range(rows, world, rank)
        part = torch.full((33,), 9.0, dtype=torch.float32, device="cuda")
        gpu_ctx.icp_accumulate_rows(Rg, tg, synth.INTR, 0, 0.1, 0.6, *args, rows, cols, r0, r1, part.data_ptr())
        total += part                                                            # what the all-reduce does
    torch.cuda.synchronize()
    total = total.cpu().numpy()
    assert np.array_equal(total[27:], np.float32(w_counts))                      # counts: exact
    scale = np.abs(np.float32(w_sums)).max()
    assert np.abs(total[:27] - np.float32(w_sums)).max() <= 1e-4 * scale         # sums: to rounding
    o_sums, o_counts = oracle.icp_accumulate(Rg.reshape(3, 3), tg, synth.INTR, 0, 0.1, 0.6, vcurr, ncurr, vprev, nprev)
    assert np.array_equal(o_counts, np.uint64(w_counts))
    assert np.abs(total[:27] - o_sums).max() <= 1e-4 * np.abs(o_sums).max()

    # an empty band gives zeros; a bad band is refused
    part = torch.full((33,), 9.0, dtype=torch.float32, device="cuda")
    gpu_ctx.icp_accumulate_rows(Rg, tg, synth.INTR, 0, 0.1, 0.6, *args, rows, cols, 17, 17, part.data_ptr())
    torch.cuda.synchronize()
    assert (part == 0).all().item()
    with pytest.raises(capi.YakHipError):
        gpu_ctx.icp_accumulate_rows(Rg, tg, synth.INTR, 0, 0.1, 0.6, *args, rows, cols, 10, rows + 1, part.data_ptr())


@pytest.mark.parametrize("world", [2, 3])
def test_slab_raycasts_over_arbitrary_voxel_words_composite_to_the_whole(gpu_ctx, world):
    """A volume that did not come from integrate: every sample pair may end the march, so every slab finds candidates
    in its own layers and many winners defer their normal to a neighbour.  The composite must be the whole-volume ray
    cast and the oracle's (NaN where they have NaN, the same bits elsewhere)."""
    from test_mc_gpu import arbitrary_words

    dims, vs, tr = synth.volume_geometry((48, 40, 56))
    vol = arbitrary_words(dims, 77)
    rows, cols, intr = 90, 120, (110.0, 110.0, 59.5, 44.5)
    _, R_v2c, t_v2c = synth.scene_s2(200)
    R, t = cam2vol(R_v2c, t_v2c)
    Rinv = np.ascontiguousarray(R.T)
    halo = capi.raycast_halo(capi.volume_desc(256, dims, vs, tr, synth.MAX_WEIGHT))
    parts = []
    for rank in range(world):
        own = slab_range(dims[2], world, rank)
        img, nrm, keys, slab = slab_raycast(gpu_ctx, vol, dims, vs, tr, own, halo, R, t, Rinv, intr, rows, cols)
        w_img, w_nrm, w_keys = oracle.raycast_slab(vol, dims, vs, tr, R, t, Rinv, intr, rows, cols, own[0], own[1], nthreads=8)
        k = keys.cpu().numpy()
        assert np.array_equal(k & ~1, w_keys & ~1) and ((k == KEY_NONE) == (w_keys == KEY_NONE)).all()
        parts.append((img, nrm, keys, slab))
    c_img, c_nrm, best, n_odd = composite_on_one_gpu(gpu_ctx, parts, False, Rinv)
    whole_p, whole_n = gpu_raycast(gpu_ctx, vol, dims, vs, tr, R, t, Rinv, intr, rows, cols)
    o_p, o_n, _, hits = oracle.raycast(vol, dims, vs, tr, R, t, Rinv, intr, rows, cols, nthreads=8)
    assert hits > 1000
    for got, want in ((c_img, whole_p), (c_nrm, whole_n), (c_img, o_p), (c_nrm, o_n)):
        g = got.cpu().numpy()
        nan = np.isnan(want)
        assert np.array_equal(np.isnan(g), nan)
        assert np.array_equal(g[~nan].view(np.uint32), want[~nan].view(np.uint32))
