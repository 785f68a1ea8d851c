#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes: one calibration copy of known size, then the C2
pipeline (n=4096, 3 primes, symmetric, B=65536) three times."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import vectors as V
import __graft_entry__ as ge
pkg = ge.load_package()
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
n, npr, B = 4096, 3, 65536
dev = torch.device("cuda:0")
# calibration: 1 GiB read + 1 GiB write, 16 B per lane
src = torch.empty(1 << 28, dtype=torch.float32, device=dev).normal_()
dst = torch.empty_like(src)
dst.copy_(src); torch.cuda.synchronize()
ctx = pkg.Context(n, npr); ctx.reserve(B); ctx.set_secret_key(V.secret_key(n))
ctx.set_debug_flags(int(os.environ.get("SE_PMC_FLAGS", "0")))   # e.g. 2 = no redraw phase (traffic of the bulk alone)
vals = torch.from_numpy(V.bench_values(B, n)).to(dev)
ss_np, sd_np = V.bench_seeds(B)
ss, sd = torch.from_numpy(ss_np).to(dev), torch.from_numpy(sd_np).to(dev)
c0 = torch.empty((B, npr, n), dtype=torch.int32, device=dev); c1 = torch.empty_like(c0)
for _ in range(3):
    if wl == "c5":
        ctx.encode_ntt(vals, c0)
    else:
        ctx.encrypt_sym(vals, ss, sd, c0, c1)
    torch.cuda.synchronize()
print("done")
