#!/usr/bin/env python3
"""Latency of small host-pointer symmetric calls (prime speculation window), per batch size."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import vectors as V
import __graft_entry__ as ge
pkg = ge.load_package()
for (n, npr, Bs) in ((4096, 3, (16, 64, 128, 256, 512)), (16384, 6, (8, 16, 32)), (8192, 6, (4, 13, 32))):
    ctx = pkg.Context(n, npr); ctx.set_secret_key(V.secret_key(n))
    for B in Bs:
        vals = V.bench_values(B, n); ss, sd = V.bench_seeds(B)
        c0 = np.zeros((B, npr, n), np.uint32); c1 = np.zeros_like(c0)
        for _ in range(3): ctx.encrypt_sym_host(vals, ss, sd, out=(c0, c1))
        t0 = time.perf_counter()
        for _ in range(10): ctx.encrypt_sym_host(vals, ss, sd, out=(c0, c1))
        t = (time.perf_counter() - t0) / 10
        print(f"n={n} np={npr} B={B}: {t*1e3:.2f} ms per call ({B/t/1e3:.1f} k ct/s)", flush=True)
    ctx.close()
