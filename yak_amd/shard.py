"""z-slab sharding of a TSDF volume over ranks.

integrate needs no data-path collective: every rank integrates layers [z0, z1) of the same frame and
the slabs concatenate, in rank order, to the single-GPU volume bit for bit (the kernel replays the
reference's float accumulation from z = 0).  torch.distributed is used around that path (timing
reduction for the benchmark, gathering slabs for downloadTSDF) and for the two steps of the tracking
side that do exchange data: compositing the per-slab ray casts (composite_raycast) and adding the
per-rank ICP sums (allreduce_icp_sums).
"""
import numpy as np

# global volume per world size for the weak-scaling benchmark: 512^3 voxels per rank, z kept shallow
BENCH_GLOBAL_DIMS = {1: (512, 512, 512), 2: (1024, 512, 512), 4: (1024, 1024, 512), 8: (1024, 1024, 1024)}


def slab_range(z_layers, world, rank):
    """Layers [z0, z1) owned by `rank`: contiguous, in rank order, sizes differing by at most one."""
    if not 0 <= rank < world:
        raise ValueError("rank %d outside world %d" % (rank, world))
    base, extra = divmod(int(z_layers), int(world))
    z0 = rank * base + min(rank, extra)
    return z0, z0 + base + (1 if rank < extra else 0)


def reduce_timing(dist, elapsed_s, units, device="cpu"):
    """Whole-job figures from per-rank ones: MAX of the elapsed time, SUM of the units processed."""
    import torch

    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    u = torch.tensor([float(units)], dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())


def gather_volume(dist, slab, dims, dst=0):
    """downloadTSDF for a sharded volume: slabs (int32/uint32 tensors, layers in z order) are gathered
    on rank `dst` and concatenated; returns the whole volume there, None elsewhere."""
    import torch

    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    if world == 1:
        return slab
    rank = dist.get_rank()
    plane = int(dims[0]) * int(dims[1])
    sizes = [(slab_range(dims[2], world, r)[1] - slab_range(dims[2], world, r)[0]) * plane for r in range(world)]
    assert slab.numel() == sizes[rank]
    pad = max(sizes)
    buf = torch.zeros(pad, dtype=slab.dtype, device=slab.device)
    buf[:slab.numel()] = slab.reshape(-1)
    out = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst)
    if rank != dst:
        return None
    return torch.cat([o[:n] for o, n in zip(out, sizes)])


def held_range(z_layers, z0, z1, halo):
    """Layers a rank must HOLD to ray-cast its own layers [z0, z1): the halo on either side, cut at the
    volume's ends.  The halo layers are integrated redundantly (integrate is a per-voxel function of the
    frame), never exchanged."""
    return max(0, int(z0) - int(halo)), min(int(z_layers), int(z1) + int(halo))


KEY_NONE = 0x7FFFFFFF


def _as_bits(img):
    import torch

    return img.view(torch.int32) if img.dtype == torch.float32 else img.to(torch.int32)


def _from_bits(bits, like):
    import torch

    return bits.view(torch.float32) if like.dtype == torch.float32 else bits.to(like.dtype)


def composite_raycast(dist, keys, images, fill, resolve=None, all_reduce=None):
    """Per-slab ray casts -> the whole-volume image (SURVEY.md 8e option 1), on every rank.

    keys: int32 tensor (rows, cols) from yakhip_raycast_slab; images: the payload tensors it wrote
    (float32 (rows, cols, 4) points / normals, int16 (rows, cols) depth); fill: what a pixel without any
    march-ending pair holds, per image (NaN for the maps, 0 for depth).  The slab with the smallest key
    owns the pixel; keys of different slabs never tie.  Two collectives: MIN over the keys, then a SUM of
    the payload bit patterns with every non-winner zeroed (one non-zero term: exact).

    resolve(keys, images): the rank's yakhip_raycast_resolve on the composited tensors, in place.  It is
    called when some winning key is odd (a hit whose normal has to be computed by the slab that holds the
    refined point); the pixels concerned -- the same list on every rank -- are then composited once more.
    all_reduce(dist, tensor, "sum" | "min"): replaces dist.all_reduce (e.g. to stage device tensors through
    the host for a CPU-only backend).
    Returns (composited images, best keys, number of pixels still unresolved: 0 unless resolve is None)."""
    import torch

    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    if all_reduce is None:
        def all_reduce(d, tensor, op):
            d.all_reduce(tensor, op=d.ReduceOp.SUM if op == "sum" else d.ReduceOp.MIN)
            return tensor
    best = keys.clone()
    if world > 1:
        all_reduce(dist, best, "min")
    mine = (keys == best) & (best != KEY_NONE)
    none = best == KEY_NONE
    out = []
    for img, f in zip(images, fill):
        bits = _as_bits(img)
        m = mine if bits.dim() == 2 else mine.unsqueeze(-1)
        bits = torch.where(m, bits, torch.zeros_like(bits))
        if world > 1:
            all_reduce(dist, bits, "sum")
        res = _from_bits(bits, img)
        n = none if res.dim() == 2 else none.unsqueeze(-1)
        out.append(torch.where(n, torch.full_like(res, f), res).contiguous())

    odd = (best != KEY_NONE) & ((best & 1) == 1)
    idx = odd.reshape(-1).nonzero().squeeze(1)  # the same pixels on every rank
    if idx.numel() == 0 or resolve is None:
        return out, best, int(idx.numel())
    keys2 = best.clone()
    resolve(keys2, out)
    k2 = keys2.reshape(-1)[idx]
    b2 = k2.clone()
    if world > 1:
        all_reduce(dist, b2, "min")
    done = (b2 & 1) == 0
    mine2 = (k2 == b2) & done
    n_pix = best.numel()
    for o in out:
        flat = o.reshape(n_pix, -1)
        old = _as_bits(flat[idx])
        bits = torch.where(mine2.unsqueeze(-1), old, torch.zeros_like(old))
        if world > 1:
            all_reduce(dist, bits, "sum")
        flat[idx] = _from_bits(torch.where(done.unsqueeze(-1), bits, old), o)
    best.reshape(-1)[idx] = b2
    return out, best, int((~done).sum().item())


def row_range(rows, world, rank):
    """Image rows [r0, r1) of one pyramid level that `rank` accumulates ICP sums over."""
    return slab_range(rows, world, rank)


def allreduce_icp_sums(dist, sums33):
    """Adds the 33 per-rank values of yakhip_icp_accumulate_rows over the ranks, in place (float32 tensor
    on the rank's device): every rank then holds the sums of the whole image and solves the same 6 x 6
    system.  19 collectives of 132 bytes per frame: latency-bound."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(sums33, op=dist.ReduceOp.SUM)
    return sums33
