"""Batch sharding across the GPUs of one node (one process per GPU, torch.distributed).

Every plaintext -> ciphertext unit is independent (own seeds; sk / pk / tables are replicated
read-only per GPU), so the batch is cut into contiguous blocks of the batch index and NO
collective runs on the data path.  The only cross-GPU step the path has is the optional final
gather of ciphertext records; it is a plain gather/all_gather of fixed-size
records in rank order, which by construction reproduces the single-process record order.
"""


def shard_bounds(total, rank, world):
    """Contiguous block [lo, hi) of rank `rank`; blocks differ in size by at most one unit."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_records(local, dist, dst=0, chunk_records=None):
    """Gather per-rank record slabs [B_r, ...] to rank `dst` in rank order.

    Returns the concatenated tensor on dst, None elsewhere.  Works for equal or unequal shard
    sizes (sizes are exchanged first).  `chunk_records` bounds the size of each collective.
    """
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device))
    sizes = [int(s.item()) for s in sizes]
    maxb = max(sizes)
    step = chunk_records or maxb or 1
    out = None
    if rank == dst:
        out = torch.empty((sum(sizes),) + tuple(local.shape[1:]), dtype=local.dtype,
                          device=local.device)
    offs = [sum(sizes[:r]) for r in range(world)]
    for start in range(0, maxb, step):
        n_here = max(0, min(step, local.shape[0] - start))
        pad = torch.zeros((step,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        if n_here:
            pad[:n_here] = local[start:start + n_here]
        bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
        dist.gather(pad, bufs, dst=dst)
        if rank == dst:
            for r in range(world):
                k = max(0, min(step, sizes[r] - start))
                if k:
                    out[offs[r] + start:offs[r] + start + k] = bufs[r][:k]
    return out
