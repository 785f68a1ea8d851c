#!/usr/bin/env python3
"""Timing ablations of the uniform sampler (cdna guide: ablate before optimising).
flags: 2 = no phase 2 (the store / bookkeeping ablations 1 and 4 went away with the branch-free
emit: they cost instructions in the hot loop)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import vectors as V
import __graft_entry__ as ge
pkg = ge.load_package()
n, npr = 4096, 3
dev = torch.device("cuda:0")
for B in (65536, 131072, 262144):
    ctx = pkg.Context(n, npr)
    ctx.reserve(B)
    seeds = torch.from_numpy(V.derive_seeds("abl", 1024)).to(dev).repeat(B // 1024, 1).contiguous()
    out = torch.empty((B, npr, n), dtype=torch.int32, device=dev)
    for flags in (0, 2):
        ctx.set_debug_flags(flags)
        ctx.sample_uniform(seeds, out); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ctx.sample_uniform(seeds, out); ctx.sample_uniform(seeds, out); b.record(); torch.cuda.synchronize()
        print(f"B={B} flags={flags}: {a.elapsed_time(b)/2:.3f} ms", flush=True)
    ctx.close(); del out
