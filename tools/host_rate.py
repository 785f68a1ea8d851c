#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry point (never the bench `value`)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import vectors as V
import __graft_entry__ as ge
pkg = ge.load_package()
n, npr, B = 4096, 3, 16384
ctx = pkg.Context(n, npr); ctx.set_secret_key(V.secret_key(n))
vals = V.bench_values(B, n); ss, sd = V.bench_seeds(B)
ctx.encrypt_sym_host(vals[:64], ss[:64], sd[:64])
for _ in range(2):
    t0 = time.perf_counter(); r = ctx.encrypt_sym_host(vals, ss, sd); t = time.perf_counter() - t0
    print(f"host-pointer entry, B={B}: {t*1e3:.1f} ms = {B/t/1e3:.1f} k ct/s ({B*(8192+128+98304)/t/1e9:.2f} GB/s over PCIe incl. allocation and pageable copies)")
