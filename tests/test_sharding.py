"""The N>1 path on CPU: world_size-2 gloo processes shard a batch, each produces its block of
ciphertext records (with the CPU oracle standing in for the GPU kernels -- the checker, not the
product), rank 0 gathers, and the result must equal the single-process record order."""
import os
import socket

import numpy as np
import pytest

import vectors as V


def test_shard_bounds_cover_and_balance():
    import __graft_entry__ as ge
    ge.load_package()
    from seal_embedded_amd.sharding import shard_bounds
    for total in (0, 1, 7, 8, 65536, 262144 + 3):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(8, 2, 2)


def _worker(rank, world, port, total, q):
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    ge.load_package()
    from seal_embedded_amd.sharding import gather_records, shard_bounds
    from oracle.pyoracle import Oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, npr = 1024, 1
    lo, hi = shard_bounds(total, rank, world)
    o = Oracle(n, npr)
    sk = V.secret_key(n)
    vals = V.bench_values(hi - lo, n, first=lo)
    ss, sd = V.bench_seeds(hi - lo, first=lo)
    ok, c0, c1 = o.encrypt_sym_batch(vals, ss, sd, sk, nthreads=1)
    rec = torch.from_numpy(np.stack([c0, c1], axis=1).view(np.int32))
    out = gather_records(rec, dist, dst=0, chunk_records=2)
    if rank == 0:
        q.put(out.numpy().view(np.uint32))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_gather_matches_single_process():
    import torch.multiprocessing as mp
    from oracle.pyoracle import Oracle
    total, world = 7, 2          # uneven shards: 4 + 3
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n, npr = 1024, 1
    o = Oracle(n, npr)
    ok, c0, c1 = o.encrypt_sym_batch(V.bench_values(total, n), *V.bench_seeds(total),
                                     V.secret_key(n), nthreads=1)
    assert got.shape == (total, 2, npr, n)
    assert (got[:, 0] == c0).all() and (got[:, 1] == c1).all()


def test_bench_contract_constants():
    """bench.py's workload table against SURVEY.md 8(d): algorithmic bytes per unit, batch sizes of
    the BASELINE configs, and the JSON keys the driver parses (static checks: no GPU here)."""
    import ast
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "bench.py")).read()
    tree = ast.parse(src)
    wl = None
    for node in tree.body:
        if isinstance(node, ast.Assign) and getattr(node.targets[0], "id", "") == "WORKLOADS":
            wl = eval(compile(ast.Expression(node.value), "bench.py", "eval"))
    assert wl is not None
    assert wl["c2"] == (4096, 3, "sym", 65536, 106624)
    assert wl["c3"] == (4096, 3, "asym", 65536, 106560)
    assert wl["c4"] == (16384, 6, "sym", 32768, 819328)
    assert wl["c5"] == (4096, 3, "encode", 262144, 57344)
    assert wl["c1"] == (1024, 1, "sym", 1, 10368)
    for key in ('"metric"', '"value"', '"unit"', '"n_gpus"', '"steps"', '"warmup"', '"ms_per_step"',
                '"higher_is_better"', '"scaling"', '"vs_baseline"', '"dtype"', '"data"', '"config"',
                '"roofline"', '"cpu_baseline"', '"bound"', '"achieved"', '"peak"', '"frac"', '"traffic"',
                '"cores"', '"kind"', '"sample"'):
        assert key in src, key
