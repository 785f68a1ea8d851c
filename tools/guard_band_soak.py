#!/usr/bin/env python3
"""Adversarial soak for the guard band of the half-size IFFT (encode_encrypt.hip, encode_pair_half): plaintexts whose
coefficients sit ON or within rounding noise OF a half-integer -- constant slot vectors (exact ties), sparse slot vectors
v = (2k+1) / 2^15 at a few slots (m_j = (k + 0.5) cos(phi_j): ties decided by the reference's own rounding errors),
small-integer slot vectors, and mixtures -- encoded through the fused kernel (pair form) and compared, record for record,
with the threaded C oracle (encode + RNS + NTT = BASELINE config 5's path).  GPU box only; prints one line.
    python tools/guard_band_soak.py [plaintexts=40000] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
from __graft_entry__ import load_package
from oracle import pyoracle
from oracle.pyoracle import Oracle
pkg = load_package()
dev = torch.device("cuda:0")
n, npr = 4096, 3
total = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.default_rng(seed)
o = Oracle(n, npr)
ctx = pkg.Context(n, npr, 0)
nt = pyoracle.host_threads()
done, chunk, t0 = 0, 4096, time.time()
while done < total:
    B = min(chunk, total - done)
    vals = np.zeros((B, n // 2), dtype=np.float32)
    kind = rng.integers(0, 5, B)
    for b in range(B):
        k = kind[b]
        if k == 0:      # exact tie at coefficient 0 (constant vector), either sign
            vals[b, :] = np.float32((2 * int(rng.integers(0, 1 << 20)) + 1) / 2.0 ** 26) * (1 if rng.integers(2) else -1)
        elif k == 1:    # 1..3 non-zero slots: ties decided by rounding noise
            for _ in range(int(rng.integers(1, 4))):
                vals[b, int(rng.integers(0, n // 2))] = np.float32((2 * int(rng.integers(0, 1 << 12)) + 1) / 2.0 ** 15)
        elif k == 2:    # small integers / 2^15 everywhere: coefficients on a 2^-? lattice
            vals[b, :] = (rng.integers(-64, 65, n // 2) / 2.0 ** 15).astype(np.float32)
        elif k == 3:    # the bench distribution plus one tie-making slot
            vals[b, :] = (rng.integers(0, 256, n // 2) / -10.0).astype(np.float32)
            vals[b, int(rng.integers(0, n // 2))] += np.float32((2 * int(rng.integers(0, 64)) + 1) / 2.0 ** 15)
        else:           # constant + sparse
            vals[b, :] = np.float32((2 * int(rng.integers(0, 1 << 10)) + 1) / 2.0 ** 26)
            vals[b, int(rng.integers(0, n // 2))] += np.float32(1.0 / 2.0 ** 15)
    out = torch.zeros((B, npr, n), dtype=torch.int32, device=dev)
    st = torch.zeros(B, dtype=torch.uint8, device=dev)
    ctx.encode_ntt(torch.from_numpy(vals).to(dev), out, status=st)
    torch.cuda.synchronize()
    ok, e = o.encode_ntt_batch(vals, nthreads=nt)
    got = out.cpu().numpy().view(np.uint32)
    if not (ok and bool(st.all()) and np.array_equal(got, e)):
        bad = np.argwhere((got != e).reshape(B, -1).any(axis=1)).ravel()
        print("GUARD-BAND MISMATCH seed", seed, "chunk at", done, "plaintexts", bad[:8].tolist(), "kinds", kind[bad[:8]].tolist())
        sys.exit(1)
    done += B
print("guard-band soak ok: %d adversarial plaintexts bit-exact in %.0f s (seed %d)" % (done, time.time() - t0, seed))
