#!/usr/bin/env python3
"""Per-step cost of the uniform sampler's squeeze (timing-only debug flags; results are WRONG with flag 2):
flags 2|8 = bulk squeeze only, wave-local geometry; 2 = bulk only with helper waves present but idle;
0 / 8 = the real kernels.  python tools/chain_step_probe.py"""
import sys, time, importlib
sys.path[:0] = ['.', 'tests']
import torch
import vectors as V
mod = importlib.import_module('seal-embedded_amd')
dev = torch.device('cuda:0')
for n, npr, B in ((16384, 6, 32768), (4096, 3, 65536), (4096, 3, 32768)):
    ctx = mod.Context(n, npr, 0)
    ss, _ = V.bench_seeds(B)
    ss = torch.from_numpy(ss).to(dev)
    out = torch.empty((B, npr, n), dtype=torch.int32, device=dev)
    ctr = torch.zeros(B, dtype=torch.int64, device=dev)
    ctx.reserve(B)
    steps = npr * ((n * 4 + 135) // 136)
    for flags in (0, 8, 2, 10):
        ctx.set_debug_flags(flags)
        for _ in range(2): ctx.sample_uniform(ss, out, ctr_out=ctr)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): ctx.sample_uniform(ss, out, ctr_out=ctr)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        print(f"n={n} np={npr} B={B} flags={flags:2d}: {ms:7.2f} ms  ({ms * 1e3 / steps:.2f} us per bulk step if squeeze only)", flush=True)
    ctx.close()
