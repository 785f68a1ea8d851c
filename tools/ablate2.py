#!/usr/bin/env python3
"""Why is k_sample_uniform slower inside the pipeline than alone?  Same seeds, same buffers."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import vectors as V
import __graft_entry__ as ge
pkg = ge.load_package()
n, npr, B = 4096, 3, 65536
dev = torch.device("cuda:0")
ctx = pkg.Context(n, npr); ctx.reserve(B); ctx.set_secret_key(V.secret_key(n))
vals = torch.from_numpy(V.bench_values(B, n)).to(dev)
ss_np, sd_np = V.bench_seeds(B)
ss, sd = torch.from_numpy(ss_np).to(dev), torch.from_numpy(sd_np).to(dev)
c0 = torch.empty((B, npr, n), dtype=torch.int32, device=dev); c1 = torch.empty_like(c0)
err = torch.empty((B, n), dtype=torch.int8, device=dev)
def timed(f, reps=3):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
print("uniform alone (bench seeds, c1 buffer): %.3f ms" % timed(lambda: ctx.sample_uniform(ss, c1)))
print("cbd alone: %.3f ms" % timed(lambda: ctx.sample_cbd(sd, err, n // 16)))
print("cbd+uniform back to back: %.3f ms" % timed(lambda: (ctx.sample_cbd(sd, err, n // 16), ctx.sample_uniform(ss, c1))))
for ov, sp in ((0, 0), (1, 0), (0, 1), (1, 1)):
    ctx.set_pipeline(ov, sp)
    print("full encrypt_sym overlap=%d split=%d: %.3f ms" % (ov, sp, timed(lambda: ctx.encrypt_sym(vals, ss, sd, c0, c1), reps=5)))
    ctx.set_profiling(True); ctx.stage_ms(True)
    for _ in range(3): ctx.encrypt_sym(vals, ss, sd, c0, c1)
    torch.cuda.synchronize(); print("   stages (ms per step):", {k: round(v[0] / 3, 3) for k, v in ctx.stage_ms(True).items() if v[1]})
    ctx.set_profiling(False)
print("uniform alone again: %.3f ms" % timed(lambda: ctx.sample_uniform(ss, c1)))
# sustained: 20 back-to-back uniform launches
print("uniform x20 sustained: %.3f ms each" % timed(lambda: ctx.sample_uniform(ss, c1), reps=20))
