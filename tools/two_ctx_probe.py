#!/usr/bin/env python3
"""Upper bound of cross-batch pipelining: consecutive batches alternate between TWO contexts (own scratch)
on two streams with separate output slabs, against one context / one stream (the serial shape of bench.py).
  python tools/two_ctx_probe.py [c2|c3|c5] [steps]"""
import sys, time, importlib
sys.path[:0] = ['.', 'tests']
import torch
import bench
import vectors as V
mod = importlib.import_module('seal-embedded_amd')
w = sys.argv[1] if len(sys.argv) > 1 else 'c2'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n, npr, mode, B = bench.WORKLOADS[w]
dev = torch.device('cuda:0')
ctxs = [mod.Context(n, npr, 0) for _ in range(2)]
sk = V.secret_key(n)
for c in ctxs:
    if mode == 'sym':
        c.set_secret_key(sk)
    elif mode == 'asym':
        pk0, pk1 = c.gen_public_key(sk, bytes(64), bytes(range(64)))
        c.set_public_key(pk0, pk1)
    c.reserve(B)
vals = bench.bench_values_device(B, n, dev)
ss, sd = V.bench_seeds(B) if mode != 'encode' else (None, None)
if ss is not None:
    ss, sd = torch.from_numpy(ss).to(dev), torch.from_numpy(sd).to(dev)
outs = [(torch.empty((B, npr, n), dtype=torch.int32, device=dev),
         torch.empty((B, npr, n), dtype=torch.int32, device=dev) if mode != 'encode' else None) for _ in range(2)]
st = [torch.zeros(B, dtype=torch.uint8, device=dev) for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]

def step(i, k):
    c, (c0, c1), s = ctxs[i], outs[k], st[k]
    if mode == 'sym':
        c.encrypt_sym(vals, ss, sd, c0, c1, status=s)
    elif mode == 'asym':
        c.encrypt_asym(vals, sd, c0, c1, status=s)
    else:
        c.encode_ntt(vals, c0, status=s)

def run(two):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        i = k & 1 if two else 0
        with torch.cuda.stream(streams[i]):
            step(i, k & 1)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

for _ in range(2):
    run(False), run(True)
for rep in range(2):
    print(w, 'one context  %.3f ms/step' % run(False), '  two contexts alternating %.3f ms/step' % run(True), flush=True)
ref = [o.clone() for o in outs[0] if o is not None]
run(False)
assert all(torch.equal(a, b) for a, b in zip(ref, [o for o in outs[0] if o is not None]))
print('outputs identical; status', bool(st[0].all()), bool(st[1].all()))
