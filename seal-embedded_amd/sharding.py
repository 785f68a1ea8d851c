"""Batch sharding across the GPUs of one node (one process per GPU, torch.distributed).

Every plaintext -> ciphertext unit is independent (own seeds; sk / pk / tables are replicated
read-only per GPU), so the batch is cut into contiguous blocks of the batch index and NO
collective runs on the data path.  The only cross-GPU step the path has is the optional final
gather of ciphertext records to one rank.

The gather is point-to-point: every source rank writes its block straight into its slice of the
root's output slab (`batch_isend_irecv`: one group of concurrent sends, each over the source's own
xGMI link to the root -- 7 links x ~153 GB/s into the root on an 8-GPU node, against ~153 GB/s for a
ring), in pieces of at most `chunk_bytes`.  No padding, no staging copies: records are fixed-size
and land in rank order, which by construction is the single-process record order.  The root's own
block is a local copy -- or nothing at all when the root produced it in place (`local` is already
the root's slice of `out`).
"""


def shard_bounds(total, rank, world):
    """Contiguous block [lo, hi) of rank `rank`; blocks differ in size by at most one unit."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_sizes(local_count, dist, device):
    """Records held by every rank (one tiny all_gather)."""
    import torch
    world = dist.get_world_size()
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([local_count], dtype=torch.int64, device=device))
    return [int(s.item()) for s in sizes]


def gather_records(local, dist, dst=0, out=None, chunk_bytes=1 << 30, sizes=None):
    """Gather per-rank record slabs [B_r, ...] to rank `dst` in rank order.

    Returns the concatenated tensor on dst (`out` if given: shape [sum B_r, ...]), None elsewhere.
    Shards may be unequal or empty.  `chunk_bytes` bounds each point-to-point message.
    """
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    if sizes is None:
        sizes = shard_sizes(local.shape[0], dist, local.device)
    if sizes[rank] != local.shape[0]:
        raise ValueError("sizes[rank] does not match the local slab")
    rec_shape = tuple(local.shape[1:])
    rec_elems = 1
    for d in rec_shape:
        rec_elems *= d
    rec_bytes = rec_elems * local.element_size()
    step = max(1, chunk_bytes // max(1, rec_bytes))          # records per message
    offs = [sum(sizes[:r]) for r in range(world)]
    total = sum(sizes)
    if rank == dst:
        if out is None:
            out = torch.empty((total,) + rec_shape, dtype=local.dtype, device=local.device)
        elif tuple(out.shape) != (total,) + rec_shape or out.dtype != local.dtype:
            raise ValueError("out must be [sum of shard sizes, ...record shape] of the records' dtype")
    ops = []
    if rank == dst:
        mine = out[offs[rank]:offs[rank] + sizes[rank]]
        if sizes[rank] and mine.data_ptr() != local.data_ptr():
            mine.copy_(local)
        for r in range(world):
            if r == dst:
                continue
            for lo in range(0, sizes[r], step):
                hi = min(sizes[r], lo + step)
                ops.append(dist.P2POp(dist.irecv, out[offs[r] + lo:offs[r] + hi], r))
    else:
        loc = local.contiguous()
        for lo in range(0, sizes[rank], step):
            hi = min(sizes[rank], lo + step)
            ops.append(dist.P2POp(dist.isend, loc[lo:hi], dst))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out if rank == dst else None
