"""z-slab sharding of a TSDF volume over ranks.

integrate needs no data-path collective: every rank integrates layers [z0, z1) of the same frame and
the slabs concatenate, in rank order, to the single-GPU volume bit for bit (the kernel replays the
reference's float accumulation from z = 0).  torch.distributed is used only around the path: timing
reduction for the benchmark and gathering slabs for downloadTSDF.
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
