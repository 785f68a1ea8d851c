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
    # tiny messages (2 records each) force the chunked path; the root's own block is produced IN PLACE
    # in its slice of the gathered slab on the second pass (no local copy then)
    out = gather_records(rec, dist, dst=0, chunk_bytes=2 * 2 * npr * n * 4)
    sizes = [shard_bounds(total, r, world)[1] - shard_bounds(total, r, world)[0] for r in range(world)]
    if rank == 0:
        first = out.clone()
        slab = torch.zeros_like(out)
        slab[lo:hi] = rec
        out2 = gather_records(slab[lo:hi], dist, dst=0, out=slab, sizes=sizes)
        assert out2.data_ptr() == slab.data_ptr() and bool((out2 == first).all())
        q.put(out.numpy().view(np.uint32))
    else:
        gather_records(rec, dist, dst=0, sizes=sizes)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total,world", [(7, 2), (1, 2), (5, 3)])   # uneven: 4+3; one EMPTY shard; 2+2+1
def test_two_rank_gloo_gather_matches_single_process(total, world):
    import torch.multiprocessing as mp
    from oracle.pyoracle import Oracle
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
    import bench
    wl = bench.WORKLOADS
    assert wl["c2"] == (4096, 3, "sym", 65536) and bench.bytes_per_unit("sym", 4096, 3) == 106624
    assert wl["c3"] == (4096, 3, "asym", 65536) and bench.bytes_per_unit("asym", 4096, 3) == 106560
    assert wl["c4"] == (16384, 6, "sym", 32768) and bench.bytes_per_unit("sym", 16384, 6) == 819328
    assert wl["c5"] == (4096, 3, "encode", 1048576) and bench.bytes_per_unit("encode", 4096, 3) == 57344
    assert wl["c1"] == (1024, 1, "sym", 1) and bench.bytes_per_unit("sym", 1024, 1) == 10368
    # every kernel's own bytes: the symmetric fused step reads a back once, the uniform sampler only writes it
    kb = bench.kernel_bytes_per_unit("sym", 4096, 3)
    assert kb["uniform"] == 64 + 49152 and kb["encode_encrypt"] == 8192 + 4096 + 2 * 49152
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")).read()
    for key in ('"metric"', '"value"', '"unit"', '"n_gpus"', '"steps"', '"warmup"', '"ms_per_step"',
                '"higher_is_better"', '"scaling"', '"vs_baseline"', '"dtype"', '"data"', '"config"',
                '"roofline"', '"cpu_baseline"', '"bound"', '"achieved"', '"peak"', '"frac"', '"traffic"',
                '"cores"', '"kind"', '"sample"', '"other_configs"', '"valu"', '"floor_ms_weighted"', '"frac_weighted"',
                '"gather_verified"', '"per_source_GB/s"', '"per_thread_value"'):
        assert key in src, key
    assert len(bench.kernel_source_hash()) == 16
    # the further configurations as the LAST key of the line, whole inside the last 600 bytes of stdout (the driver's
    # record keeps the tail of stdout verbatim: VERDICT r5 item 4)
    import json
    fake = [{"value": 7734567.891, "ms_per_step": 8.47312345, "roofline": {"frac": 0.10312345, "dominant_kernel":
             {"ms_per_step": 4.7312345}}, "cpu_baseline": {"value": 1.0}} for _ in range(4)]
    fake[3] = {"config": {"workload": "x"}, "error": "RuntimeError('boom " + "x" * 200 + "')"}
    line = {"metric": "m", "value": 1.0, "other_configs": fake}
    line["other_configs_summary"] = bench.configs_summary(["c3", "c4", "c5", "c1"], fake)
    text = json.dumps(line)
    tail = text[-600:]
    assert list(line)[-1] == "other_configs_summary" and '"other_configs_summary": {"c3"' in tail
    assert json.loads(tail[tail.index('{"c3"'):-1])["c5"] == {"value": 7735000.0, "ms_per_step": 8.473, "frac": 0.1031,
                                                              "dominant_ms": 4.731}
    assert "line.pop(\"other_configs_summary\", None)" in src      # re-appended after every further config: stays last
    # optional workloads beyond BASELINE.json carry the same per-unit byte formulas
    assert wl["x1"][:3] == (16384, 6, "asym") and wl["x2"][:3] == (8192, 6, "sym")
    # the opcode-weighted VALU bound: profiles/valu_mix.json is stamped with the kernel sources it was built from and
    # covers every kernel of the BASELINE workloads
    vm = bench.load_valu_mix(bench.kernel_source_hash())
    if not vm:      # stale evidence is not a correctness failure: bench.py prints null for these fields then
        pytest.skip("profiles/valu_mix.json was built for other kernel sources: python tools/valu_mix.py profiles/r04_ubench2.txt")
    for kern, mode, logn in (("k_sample_uniform", "sym", 12), ("k_sample_cbd", "sym", 12), ("k_encode_encrypt", "sym", 12),
                             ("k_encode_encrypt", "asym", 12), ("k_encode_encrypt", "encode", 12),
                             ("k_bulk_pair", "sym", 14), ("k_candidates", "sym", 14), ("k_ntt_fuse", "sym", 14),
                             ("k_encode_rns", "sym", 14), ("k_sample_ternary", "asym", 12)):
        cpi = bench.kernel_cpi(vm, kern, mode, logn)
        assert cpi is not None and 2.2 < cpi <= 4.2, (kern, cpi)


def _run_bench(world, extra, tmp_path):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SE_BENCH_STUB="stub_context", PYTHONPATH=os.path.join(root, "tests"))
    env.setdefault("SE_BENCH_CPU_BUDGET_S", "0.4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
           "--gpus", str(world)] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout        # ONE JSON line, from rank 0 only
    return json.loads(lines[0])


def test_bench_rank_logic_two_ranks_gloo(tmp_path):
    """bench.py itself under torch.distributed.run with 2 ranks (gloo, CPU tensors, the oracle-backed
    stub context of tests/stub_context.py in place of the library): rank r encrypts block r of the batch
    index (`first = rank * B`), the line reports n_gpus = 2 with the whole-job rate, and the gathered
    slab on rank 0 is bit-identical to the single-process order (checked inside the run)."""
    d = _run_bench(2, ["--steps", "2", "--warmup", "1", "--workload", "c1", "--batch", "3", "--others", "none"],
                   tmp_path)
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["batch_per_gpu"] == 3 and d["config"]["global_batch"] == 6
    assert abs(d["value"] - 6 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-6     # units of ALL ranks / max time
    assert d["vs_baseline"] is None and d["higher_is_better"] is True
    # N > 1 lines carry the CPU reference too (north_star: "in the same run"), timed by rank 0 after the gather
    cb = d["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] in ("reference", "port") and "host thread" in cb["sample"]
    g = d["gather"]
    assert g["form"] == "full" and g["bytes_into_root"] == 2 * 3 * 4 * 1024 and g["value_with_gather"] < d["value"]
    # the gathered slab was CHECKED on the root (records of every rank's block re-encrypted locally)
    assert g["gather_verified"] is True and g["verified_records"] == 2 * 3 and "first_mismatch" not in g
    assert g["per_source_GB/s"][0] is None and g["per_source_GB/s"][1] > 0
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["traffic"] is None
    assert d["ranks"]["world_size"] == 2 and d["ranks"]["backend"] == "gloo" and len(d["ranks"]["device"]) == 2


def test_bench_flags_a_gather_that_lost_bytes(tmp_path):
    """One word of the last rank's block flipped in the root's slab after the gather (test hook): the line
    still carries the measurement, `gather_verified` is false and names the record."""
    os.environ["SE_BENCH_TEST_CORRUPT_GATHER"] = "1"
    try:
        d = _run_bench(2, ["--steps", "1", "--warmup", "1", "--workload", "c1", "--batch", "4", "--others", "none",
                           "--no-cpu-baseline"], tmp_path)
    finally:
        os.environ.pop("SE_BENCH_TEST_CORRUPT_GATHER", None)
    g = d["gather"]
    assert d["value"] > 0 and g["gather_verified"] is False
    assert g["first_mismatch"] == {"record": 4 + 2, "rank": 1, "slab": "c0"}
    assert d["cpu_baseline"] is None                     # --no-cpu-baseline


def test_bench_three_ranks_seed_and_encode_forms(tmp_path):
    """Three ranks, encode-only workload (one slab): gather verified for every rank's block."""
    d = _run_bench(3, ["--steps", "1", "--warmup", "1", "--workload", "c5", "--batch", "2", "--others", "none",
                       "--no-cpu-baseline"], tmp_path)
    assert d["n_gpus"] == 3 and d["gather"]["gather_verified"] is True and d["gather"]["verified_records"] == 6


def test_bench_seed_compressed_gather_is_verified_too(tmp_path):
    """When the root cannot hold both gathered slabs (test hook) a symmetric run gathers c0 + the 64-byte shareable
    seeds; the untimed verification compares both with what the root re-encrypts itself."""
    os.environ["SE_BENCH_TEST_NO_FULL_GATHER"] = "1"
    try:
        d = _run_bench(2, ["--steps", "1", "--warmup", "1", "--workload", "c1", "--batch", "3", "--others", "none",
                           "--no-cpu-baseline"], tmp_path)
    finally:
        os.environ.pop("SE_BENCH_TEST_NO_FULL_GATHER", None)
    g = d["gather"]
    assert g["form"] == "seed-compressed" and g["gather_verified"] is True and g["verified_records"] == 6
    assert g["bytes_into_root"] == 3 * 4 * 1024 + 3 * 64


def test_launcher_with_more_ranks_than_devices(tmp_path):
    """torch.distributed.run started 2 ranks on a box that shows ONE device: rank 1 idles through the same
    collective sequence (class Collectives), rank 0 measures, the line says what happened -- and still has its
    CPU baseline (the idle rank waits for it)."""
    os.environ["SE_BENCH_STUB_DEVICES"] = "1"
    try:
        d = _run_bench(2, ["--steps", "1", "--warmup", "1", "--workload", "c1", "--batch", "2", "--others", "c1"],
                       tmp_path)
    finally:
        os.environ.pop("SE_BENCH_STUB_DEVICES", None)
    assert d["n_gpus"] == 1 and d["requested_gpus"] == 2 and "only 1 device(s)" in d["note"]
    assert d["config"]["global_batch"] == 2 and "gather" not in d
    assert d["ranks"]["ms_per_step"][1] is None and d["cpu_baseline"]["value"] > 0
    assert d["other_configs"][0]["value"] > 0


def test_bench_gathered_slab_is_single_process_order(tmp_path):
    """The same run with SE_BENCH_DUMP: rank 0 writes the gathered c0/c1 slabs; they equal the oracle's
    records for batch indices 0 .. world*B-1 in order."""
    from oracle.pyoracle import Oracle
    dump = tmp_path / "slab.npz"
    os.environ["SE_BENCH_DUMP"] = str(dump)
    try:
        _run_bench(2, ["--steps", "1", "--warmup", "1", "--workload", "c1", "--batch", "2", "--others", "none"],
                   tmp_path)
    finally:
        os.environ.pop("SE_BENCH_DUMP", None)
    got = np.load(dump)
    n, npr, total = 1024, 1, 4
    ok, c0, c1 = Oracle(n, npr).encrypt_sym_batch(V.bench_values(total, n), *V.bench_seeds(total),
                                                  V.secret_key(n), nthreads=1)
    assert (got["c0"].view(np.uint32) == c0).all() and (got["c1"].view(np.uint32) == c1).all()


def test_bench_line_survives_a_rank_lost_in_the_gather(tmp_path):
    """The contract measurement is protected: when a rank never reaches the gather (test hook
    SE_BENCH_TEST_HANG), every rank's deadline fires, rank 0 prints the line it has -- contract fields and
    roofline complete, the gather marked unfinished, "incomplete" set -- and the job exits 0."""
    os.environ["SE_BENCH_TEST_HANG"] = "gather"
    os.environ["SE_BENCH_DEADLINE_S"] = "4"
    try:
        d = _run_bench(2, ["--steps", "1", "--warmup", "1", "--workload", "c1", "--batch", "2", "--others", "none"],
                       tmp_path)
    finally:
        os.environ.pop("SE_BENCH_TEST_HANG", None)
        os.environ.pop("SE_BENCH_DEADLINE_S", None)
    assert "incomplete" in d and d["n_gpus"] == 2 and d["value"] > 0 and d["roofline"]["bound"] == "hbm"
    assert d["gather"]["form"] == "full" and "error" in d["gather"]


def _run_bench_plain(extra, tmp_path, env_extra=None):
    """`python bench.py ...` as the driver's single-GPU command line spells it: NO torch.distributed.run
    around it, no WORLD_SIZE in the environment."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SE_BENCH_STUB="stub_context", PYTHONPATH=os.path.join(root, "tests"))
    env.setdefault("SE_BENCH_CPU_BUDGET_S", "0.4")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + extra, env=env, capture_output=True,
                       text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0]), r.stderr


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` without a launcher starts its two ranks itself (torch.distributed.run on
    127.0.0.1, a free port) and prints the same ONE line: n_gpus = 2, the whole-job rate, the gather, and
    every rank's own step time next to the process group's world size."""
    d, _ = _run_bench_plain(["--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "c1", "--batch", "3",
                             "--others", "none"], tmp_path)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 6 and "requested_gpus" not in d
    assert d["ranks"]["world_size"] == 2 and len(d["ranks"]["ms_per_step"]) == 2
    assert all(t is not None and t > 0 for t in d["ranks"]["ms_per_step"])
    assert abs(max(d["ranks"]["ms_per_step"]) - d["ms_per_step"]) < 1e-9       # the line's time is the max
    assert d["gather"]["form"] == "full"


def test_bench_with_fewer_devices_than_requested(tmp_path):
    """--gpus 4 on a box that shows 2 devices, and --gpus 8 on one that shows 1: the run does not die; it
    measures on the devices there are and the line says so (n_gpus = what ran, requested_gpus, note)."""
    d, err = _run_bench_plain(["--gpus", "4", "--steps", "1", "--warmup", "1", "--workload", "c1", "--batch", "2",
                               "--others", "none"], tmp_path, {"SE_BENCH_STUB_DEVICES": "2"})
    assert d["n_gpus"] == 2 and d["requested_gpus"] == 4 and "only 2 device(s)" in d["note"]
    assert "4 requested, 2 device(s) visible" in err
    d, _ = _run_bench_plain(["--gpus", "8", "--steps", "1", "--warmup", "1", "--workload", "c1", "--batch", "2",
                             "--others", "none"], tmp_path, {"SE_BENCH_STUB_DEVICES": "1"})
    assert d["n_gpus"] == 1 and d["requested_gpus"] == 8 and d["value"] > 0 and "ranks" not in d
