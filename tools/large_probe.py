# timing of the fused kernels on plaintexts that are NOT "small" (general path) vs the normal range
import sys, time, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import importlib
mod = importlib.import_module('seal-embedded_amd')
import vectors as V
import numpy as np
dev = torch.device('cuda:0')
n, npr, B = 4096, 3, 65536
ctx = mod.Context(n, npr, 0)
ctx.set_secret_key(V.secret_key(n))
ss, sd = V.bench_seeds(B)
ss = torch.from_numpy(ss).to(dev); sd = torch.from_numpy(sd).to(dev)
c0 = torch.empty((B, npr, n), dtype=torch.int32, device=dev); c1 = torch.empty_like(c0)
st = torch.zeros(B, dtype=torch.uint8, device=dev)
ctx.reserve(B)
for name, scale in (("normal", 1.0), ("large", 100.0), ("mixed 1%", None)):
    vals = (torch.rand((B, n // 2), device=dev) * -25.5)
    if scale is None:
        vals[::100] *= 100.0
    else:
        vals *= scale
    for _ in range(2): ctx.encrypt_sym(vals, ss, sd, c0, c1, status=st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): ctx.encrypt_sym(vals, ss, sd, c0, c1, status=st)
    torch.cuda.synchronize(); print(name, 'sym ms/step', (time.perf_counter() - t0) / 5 * 1e3, 'ok', bool(st.all()))
