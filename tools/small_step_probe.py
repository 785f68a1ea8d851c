#!/usr/bin/env python3
"""Is a batch-of-one step host-bound?  Device-resident symmetric encrypt of ONE ciphertext (C1: n = 1024 x 1, and
n = 4096 x 3), 300 calls enqueued back to back: host time to enqueue a step vs time per step once the device has
drained, and the latency of a call that is synchronised every time.  Round 6 (MI355X): 50 us enqueue vs 105 us per
step at C1 -- the device, not the host, sets the pace; 159 vs 420 us at n = 4096 x 3; synchronised calls 133 / 416 us.
GPU box only:  python tools/small_step_probe.py [n nprimes]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch, numpy as np
from __graft_entry__ import load_package
pkg = load_package()
dev = torch.device("cuda:0")
shapes = [(1024, 1), (4096, 3), (16384, 6)]
if len(sys.argv) > 2:
    shapes = [(int(sys.argv[1]), int(sys.argv[2]))]
for (n, npr) in shapes:
    ctx = pkg.Context(n, npr, 0)
    B = 1
    g = torch.Generator(device="cpu"); g.manual_seed(1)
    vals = (torch.randint(0, 256, (B, n // 2), generator=g).float() / -10.0).to(dev)
    ss = torch.randint(0, 256, (B, 64), dtype=torch.uint8, generator=g).to(dev)
    sd = torch.randint(0, 256, (B, 64), dtype=torch.uint8, generator=g).to(dev)
    import vectors as V
    ctx.set_secret_key(V.secret_key(n))
    c0 = torch.empty((B, npr, n), dtype=torch.int32, device=dev); c1 = torch.empty_like(c0)
    st = torch.zeros(B, dtype=torch.uint8, device=dev)
    for _ in range(20): ctx.encrypt_sym(vals, ss, sd, c0, c1, status=st)
    torch.cuda.synchronize()
    K = 300
    t0 = time.perf_counter()
    for _ in range(K): ctx.encrypt_sym(vals, ss, sd, c0, c1, status=st)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("n=%d np=%d: host enqueue %.1f us/step, total %.1f us/step" % (n, npr, (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6))
    # one call at a time (sync after each): the latency a caller sees
    t0 = time.perf_counter()
    for _ in range(100):
        ctx.encrypt_sym(vals, ss, sd, c0, c1, status=st); torch.cuda.synchronize()
    print("   synced single calls: %.1f us" % ((time.perf_counter() - t0) / 100 * 1e6))
