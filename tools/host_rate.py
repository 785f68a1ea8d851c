#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry point (never the bench `value`).

Pageable destinations go through the pinned staging ring + memcpy pool of se_hostpipe.cpp; pinned
destinations are written by DMA directly.  Destination buffers are allocated and touched before
the timed region (first-touch page faults are the caller's, not the library's)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch
import vectors as V
import __graft_entry__ as ge
pkg = ge.load_package()
n, npr = 4096, 3
B = int(os.environ.get("SE_HOST_RATE_B", "32768"))
ctx = pkg.Context(n, npr); ctx.set_secret_key(V.secret_key(n))
vals = V.bench_values(B, n); ss, sd = V.bench_seeds(B)
ctx.encrypt_sym_host(vals[:64], ss[:64], sd[:64])
per_ct = 8192 + 128 + 98304
def run(tag, out, **kw):
    for it in range(3):
        t0 = time.perf_counter(); r = ctx.encrypt_sym_host(vals, ss, sd, out=out, **kw); t = time.perf_counter() - t0
        assert r["failed"] == 0
        print(f"{tag} B={B} run {it}: {t*1e3:.1f} ms = {B/t/1e3:.1f} k ct/s ({B*(per_ct - (98304//2 if kw else 0))/t/1e9:.2f} GB/s over PCIe)", flush=True)
c0 = np.ones((B, npr, n), dtype=np.uint32); c1 = np.ones_like(c0)
print("copy threads:", os.environ.get("SE_AMD_HOST_THREADS", "default"))
run("pageable", (c0, c1))
run("pageable seed-compressed (c0 only)", (c0, c1), seed_compressed=True)
p0 = torch.zeros((B, npr, n), dtype=torch.int32).pin_memory(); p1 = torch.zeros((B, npr, n), dtype=torch.int32).pin_memory()
run("pinned  ", (p0.numpy().view(np.uint32), p1.numpy().view(np.uint32)))
assert (p0.numpy().view(np.uint32) == c0).all()
