#!/usr/bin/env python3
"""C4 (n=16384, 6 primes, 32768 ct/GPU): serial stage durations vs the overlapped pipeline."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import vectors as V
import __graft_entry__ as ge
pkg = ge.load_package()
dev = torch.device("cuda:0")
def timed(f, reps=3):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
n, npr = 16384, 6
for B in (int(x) for x in os.environ.get("C4_B", "32768").split(",")):
    ctx = pkg.Context(n, npr); ctx.reserve(B); ctx.set_secret_key(V.secret_key(n))
    vals = torch.from_numpy(V.bench_values(1024, n)).to(dev).repeat(B // 1024, 1).contiguous()
    ss_np, sd_np = V.bench_seeds(B)
    ss, sd = torch.from_numpy(ss_np).to(dev), torch.from_numpy(sd_np).to(dev)
    c0 = torch.empty((B, npr, n), dtype=torch.int32, device=dev); c1 = torch.empty_like(c0)
    for ov, sp, fl in ((0, 2, 0), (1, 2, 0), (0, 2, 2)):
        ctx.set_pipeline(ov, sp); ctx.set_debug_flags(fl)
        t = timed(lambda: ctx.encrypt_sym(vals, ss, sd, c0, c1))
        ctx.set_profiling(True); ctx.stage_ms(True)
        for _ in range(2): ctx.encrypt_sym(vals, ss, sd, c0, c1)
        torch.cuda.synchronize(); st = {k: round(v[0] / 2, 2) for k, v in ctx.stage_ms(True).items() if v[1]}
        ctx.set_profiling(False)
        print(f"B={B} overlap={ov} split={sp} flags={fl}: {t:.2f} ms ({B/t:.1f}k ct/s) stages {st} sum {sum(st.values()):.1f}", flush=True)
    ctx.close(); del c0, c1, vals
