"""Multi-device tests that need TWO OR MORE DISTINCT GPUs (-m gpu): collected everywhere, skipped on a
one-GPU box, run on the first 8-GPU node -- so that the first scaling run exercises distinct-device peer
copies and the RCCL point-to-point gather under pytest, not only under bench.py.

Everything with one physical device listed several times lives in test_gpu_parity.py.  Checker: the CPU
oracle (and, inside the programs under test, a single-device pass of the product itself).
Reference boundary being extended: one ciphertext per call through a host callback
(device/lib/seal_embedded.h:65, seal_embedded.c:194-203); the reference has no multi-device form.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import vectors as V

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (no CPU fallback exists)")
    ndev = torch.cuda.device_count()
    buses = set()
    for d in range(ndev):
        pr = torch.cuda.get_device_properties(d)
        buses.add((pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id))
    if len(buses) < 2:
        pytest.skip(f"needs >= 2 distinct GPUs, this box shows {len(buses)}")
    import __graft_entry__ as ge
    pkg = ge.load_package()
    from oracle import pyoracle
    pyoracle.build(ref=False)
    return dict(torch=torch, pkg=pkg, ndev=ndev)


def _oracle_records(n, npr, B, first, sk, mode, pk=None):
    from oracle.pyoracle import Oracle
    o = Oracle(n, npr)
    vals = V.bench_values(B, n, first=first)
    ss, sd = V.bench_seeds(B, first=first)
    if mode == "sym":
        ok, c0, c1 = o.encrypt_sym_batch(vals, ss, sd, sk)
    else:
        ok, c0, c1 = o.encrypt_asym_batch(vals, sd, pk[0], pk[1])
    assert ok
    return vals, ss, sd, c0, c1


@pytest.mark.parametrize("mode,root", [("sym", 0), ("sym", -2), ("asym", 1), ("sym_seeded", 0)])
def test_group_over_distinct_devices_vs_oracle(env, mode, root):
    """se_amd_group over EVERY visible device: block i resident on device i, peer-to-peer gather into the root's
    slab (root 0, the last device, device 1; full and seed-compressed form), compared with the oracle record by
    record; the resident blocks on the other devices are compared too."""
    torch = env["torch"]
    ndev = env["ndev"]
    n, npr = 4096, 3
    B = 37 * ndev + 3                                     # unequal blocks
    sk = V.secret_key(n, seed=5)
    from oracle.pyoracle import Oracle
    pk = Oracle(n, npr).gen_pk(sk, bytes(64), bytes(range(64))) if mode == "asym" else None
    vals, ss, sd, e0, e1 = _oracle_records(n, npr, B, 11, sk, "asym" if mode == "asym" else "sym", pk)
    g = env["pkg"].Group(n, npr, list(range(ndev)))
    assert g.size == ndev and g.devices == list(range(ndev))
    if mode == "asym":
        g.set_public_key(*pk)
    else:
        g.set_secret_key(sk)
    first, count = g.partition(B)
    assert sum(count) == B
    root = root % ndev
    rdev = torch.device("cuda", root)
    c0_all = torch.full((B, npr, n), -1, dtype=torch.int32, device=rdev)
    c1_all = torch.full((B, npr, n), -1, dtype=torch.int32, device=rdev) if mode != "sym_seeded" else None
    bv, bss, bsd, b0, b1, bst = [], [], [], [], [], []
    for i in range(ndev):
        d = torch.device("cuda", i)
        lo, hi = first[i], first[i] + count[i]
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a[lo:hi])).to(d)
        bv.append(t(vals)), bss.append(t(ss)), bsd.append(t(sd))
        if i == root:                                     # the root produces its block in place inside the slab
            b0.append(c0_all[lo:hi])
            b1.append(c1_all[lo:hi] if c1_all is not None else None)
        else:
            b0.append(torch.zeros((count[i], npr, n), dtype=torch.int32, device=d))
            b1.append(torch.zeros((count[i], npr, n), dtype=torch.int32, device=d) if mode != "sym_seeded" else None)
        bst.append(torch.zeros(max(count[i], 1), dtype=torch.uint8, device=d))
    if mode == "asym":
        g.encrypt_asym(B, bv, bsd, b0, b1, status=bst, gather_root=root, c0_all=c0_all, c1_all=c1_all)
    elif mode == "sym":
        g.encrypt_sym(B, bv, bss, bsd, b0, b1, status=bst, gather_root=root, c0_all=c0_all, c1_all=c1_all)
    else:
        g.encrypt_sym(B, bv, bss, bsd, b0, None, status=bst, gather_root=root, c0_all=c0_all, c1_all=None)
    for i in range(ndev):
        torch.cuda.synchronize(i)
    assert (c0_all.cpu().numpy().view(np.uint32) == e0).all()
    if c1_all is not None:
        assert (c1_all.cpu().numpy().view(np.uint32) == e1).all()
    for i in range(ndev):
        lo, hi = first[i], first[i] + count[i]
        assert (b0[i].cpu().numpy().view(np.uint32) == e0[lo:hi]).all()
        assert bool(bst[i][:count[i]].all())
    g.close()


def test_c_program_over_all_devices(env, tmp_path):
    """examples/multi_device_encrypt.c without a device list = all visible devices: the program itself compares
    the peer-gathered slab with a single-device pass (gather_verified=1), and the digest equals the oracle's."""
    from oracle import pyoracle
    from oracle.pyoracle import Oracle
    import test_gpu_parity as P
    n, npr, B = 1024, 1, 16 * env["ndev"] + 5
    exe = P._build_example("multi_device_encrypt", tmp_path, hip=True)
    sk = V.secret_key(n)
    skf = tmp_path / "sk.dat"
    sk.tofile(skf)
    devs = ",".join(str(i) for i in range(env["ndev"]))
    res = subprocess.run([str(exe), str(n), str(npr), str(B), devs, str(skf)], check=True, capture_output=True,
                         text=True, timeout=600)
    kv = dict(f.split("=") for f in [l for l in res.stdout.splitlines() if l.startswith("failed=")][-1].split())
    assert kv["failed"] == "0" and kv["gather_verified"] == "1"
    assert int(kv["distinct_devices"]) == env["ndev"] >= 2
    o = Oracle(n, npr)
    h = 0xcbf29ce484222325
    for b in range(B):
        i = np.arange(n // 2, dtype=np.uint64) + np.uint64(b)
        with np.errstate(over="ignore"):
            v = ((i * np.uint64(2654435761)) % np.uint64(100000)).astype(np.float64) / 1000 - 50
        share = bytes((k + b) & 255 for k in range(64))
        seed = bytes((255 - k + 3 * b) & 255 for k in range(64))
        r = o.encrypt_sym(v.astype(np.float32), share, seed, sk)
        for j in range(npr):
            h = pyoracle.fnv1a64(r["c0"][j].tobytes(), h)
            h = pyoracle.fnv1a64(r["c1"][j].tobytes(), h)
    assert kv["all"] == "%016x" % h


def test_bench_ranks_over_rccl(env, tmp_path):
    """bench.py as the driver launches it for N = 2 (torch.distributed.run, one rank per GPU, RCCL): the line
    carries both ranks' distinct devices, the gather ran over RCCL point-to-point and was VERIFIED on the root,
    and the CPU baseline is in the N > 1 line."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    envp = dict(os.environ, SE_BENCH_CPU_BUDGET_S="1.0", SE_BENCH_NO_CLOCK="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "1", "--workload", "c2", "--batch", "4096", "--others", "none"]
    r = subprocess.run(cmd, env=envp, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["ranks"]["backend"] == "nccl" and d["ranks"]["world_size"] == 2
    devs = d["ranks"]["device"]
    assert len(set(x.split("pci=")[1] for x in devs)) == 2, devs
    g = d["gather"]
    assert g["form"] == "full" and g["gather_verified"] is True and g["per_source_GB/s"][1] > 0
    assert d["cpu_baseline"]["value"] > 0 and "incomplete" not in d
