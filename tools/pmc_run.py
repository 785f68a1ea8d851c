#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes: one calibration copy of known size, then STEPS steps of
the named BASELINE workload at its bench batch (bench.py's own configuration table)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import vectors as V
import bench
import __graft_entry__ as ge
pkg = ge.load_package()
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
STEPS = 3
n, npr, mode, B = bench.WORKLOADS[wl]
B = int(os.environ.get("SE_PMC_BATCH", B))
dev = torch.device("cuda:0")
# calibration: 1 GiB read + 1 GiB write, 16 B per lane
src = torch.empty(1 << 28, dtype=torch.float32, device=dev).normal_()
dst = torch.empty_like(src)
dst.copy_(src); torch.cuda.synchronize()
del src, dst
ctx = pkg.Context(n, npr); ctx.reserve(B)
sk = V.secret_key(n)
if mode == "sym":
    ctx.set_secret_key(sk)
elif mode == "asym":
    ctx.set_public_key(*ctx.gen_public_key(sk, bytes(64), bytes(range(64))))
ctx.set_debug_flags(int(os.environ.get("SE_PMC_FLAGS", "0")))   # form selection (include/seal_embedded_amd.h)
vals = bench.bench_values_device(B, n, dev)
ss_np, sd_np = V.bench_seeds(B) if mode != "encode" else (None, None)
ss = torch.from_numpy(ss_np).to(dev) if ss_np is not None else None
sd = torch.from_numpy(sd_np).to(dev) if sd_np is not None else None
c0 = torch.empty((B, npr, n), dtype=torch.int32, device=dev)
c1 = torch.empty_like(c0) if mode != "encode" else None
for _ in range(STEPS):
    if mode == "encode":
        ctx.encode_ntt(vals, c0)
    elif mode == "asym":
        ctx.encrypt_asym(vals, sd, c0, c1)
    else:
        ctx.encrypt_sym(vals, ss, sd, c0, c1)
    torch.cuda.synchronize()
print("done", wl, B, STEPS)
