"""Instance sharding over ranks: contiguous block partition, static plan replicated,
no data-path collective; the only exchange is the all-gather of the solved dq shards."""


def shard_range(total, rank, world):
    """[lo, hi) of the instances owned by `rank`; the first (total % world) ranks get one extra."""
    if world < 1 or rank < 0 or rank >= world or total < 0:
        raise ValueError("bad shard arguments")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def all_gather_dq(local_dq, total, group=None):
    """collect the per-rank dq shards into [total][n] on every rank (torch.distributed: RCCL on GPUs,
    gloo on CPU tensors).  Uneven shards are padded to the largest shard for the collective."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = local_dq.shape[1]
    sizes = [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)]
    mx = max(sizes)
    pad = torch.zeros((mx, n), dtype=local_dq.dtype, device=local_dq.device)
    pad[: sizes[rank]] = local_dq[: sizes[rank]]
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([bufs[r][: sizes[r]] for r in range(world)], dim=0)
