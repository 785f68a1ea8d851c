#!/usr/bin/env python3
"""Quick parity of the 8-points-per-thread fused kernel (debug flag 1 << 18) against the 16-point form (1 << 19)
and the oracle at n = 4096 x 3: encode-only and fused symmetric, with pte / ntt_pte / status, incl. plaintexts the
fast form declines.  GPU box only (tools, not a test: the -m gpu suite holds the permanent cases)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import vectors as V
from __graft_entry__ import load_package
from oracle.pyoracle import Oracle
pkg = load_package()
dev = torch.device("cuda:0")
n, npr, B = 4096, 3, 700
F8, F16 = 1 << 18, 1 << 19
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
vals = V.bench_values(B, n)
vals[5] *= 40.0          # not small: the general kernel
vals[77] *= 1000.0
vals[300, 3] = float("nan")
ss, sd = V.bench_seeds(B)
sk = V.secret_key(n)
res = {}
for form in (F16, F8):
    ctx = pkg.Context(n, npr, 0)
    ctx.set_debug_flags(form)
    ctx.set_secret_key(sk)
    ctx.set_pipeline(True, False)      # fused kernel whatever the batch
    out = torch.zeros((B, npr, n), dtype=torch.int32, device=dev)
    pte = torch.zeros((B, n), dtype=torch.int64, device=dev)
    st = torch.zeros(B, dtype=torch.uint8, device=dev)
    ctx.encode_ntt(t(vals), out, pte=pte, status=st)
    c0 = torch.zeros_like(out); c1 = torch.zeros_like(out); npte = torch.zeros_like(out)
    pte2 = torch.zeros_like(pte); st2 = torch.zeros_like(st)
    ctx.encrypt_sym(t(vals), t(ss), t(sd), c0, c1, npte, pte2, st2)
    torch.cuda.synchronize()
    res[form] = [x.cpu().numpy() for x in (out, pte, st, c0, c1, npte, pte2, st2)]
    ctx.close()
names = ["enc.out", "enc.pte", "enc.status", "sym.c0", "sym.c1", "sym.ntt_pte", "sym.pte", "sym.status"]
bad = 0
for nm, a, b in zip(names, res[F16], res[F8]):
    eq = np.array_equal(a, b)
    print(nm, "equal" if eq else "DIFFER", flush=True)
    if not eq:
        bad += 1
        d = np.argwhere(a != b)
        print("   first diffs:", d[:5].tolist(), "count", len(d))
o = Oracle(n, npr)
for b in (0, 5, 77, 300, B - 1):
    r = o.encrypt_sym(vals[b], ss[b].tobytes(), sd[b].tobytes(), sk)
    ok = (res[F8][3][b].view(np.uint32) == r["c0"]).all() and (res[F8][4][b].view(np.uint32) == r["c1"]).all()
    print("oracle ct", b, "ok" if ok else "MISMATCH", "status", int(res[F8][7][b]))
    bad += 0 if ok else 1
print("T8_CHECK", "PASS" if bad == 0 else "FAIL")
sys.exit(1 if bad else 0)
