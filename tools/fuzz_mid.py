#!/usr/bin/env python3
"""Randomised parity soak for MID-SIZE batches (GPU): the sizes at which the library switches between the
wave-per-ciphertext, the staged (lane pairs + candidate kernel + wave resolve) and the lane-per-ciphertext forms of
the uniform sampler on its own (no debug flags), every ciphertext of every batch compared with the threaded CPU
oracle.  FUZZ_SECONDS (default 300), FUZZ_SEED."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import vectors as V
import __graft_entry__ as ge
from oracle.pyoracle import Oracle, host_threads
pkg = ge.load_package()
dev = torch.device("cuda:0")
budget = float(os.environ.get("FUZZ_SECONDS", "300"))
master = int(os.environ.get("FUZZ_SEED", str(int(time.time()))))
print("fuzz_mid master seed", master, flush=True)
rng = random.Random(master)
thr = host_threads()
ctxs = {}
def T(a): return torch.from_numpy(np.ascontiguousarray(a)).to(dev)
t0 = time.time(); cases = cts = 0
while time.time() - t0 < budget:
    n, npr = rng.choice([(1024, 1), (2048, 1), (4096, 1), (4096, 3)])
    lim = 16 * 256
    B = rng.choice([lim - 1, lim, lim + 1, lim + 63, 2 * lim + 5, 3 * lim, 128 * 256 - 1, 128 * 256, 128 * 256 + 1,
                    rng.randrange(lim, 20000)])
    if n == 4096 and npr == 3 and B > 3 * lim:
        B = rng.choice([lim + 1, 2 * lim + 5, 3 * lim])          # oracle time
    mode = rng.choice(["sym", "sym", "asym"])
    key = (n, npr)
    if key not in ctxs:
        ctx = pkg.Context(n, npr); sk = V.secret_key(n, seed=n + npr); ctx.set_secret_key(sk)
        o = Oracle(n, npr)
        pk0, pk1 = o.gen_pk(sk, bytes(range(64)), bytes(range(1, 65)))
        ctx.set_public_key(pk0, pk1)
        ctxs[key] = (ctx, o, sk, pk0, pk1)
    ctx, o, sk, pk0, pk1 = ctxs[key]
    seed = rng.getrandbits(32)
    nr = np.random.default_rng(seed)
    vals = nr.uniform(-30, 30, (B, n // 2)).astype(np.float32)
    ss = nr.integers(0, 256, (B, 64), dtype=np.uint8); sd = nr.integers(0, 256, (B, 64), dtype=np.uint8)
    ov, sp = rng.choice([(1, 2), (1, 2), (1, 1), (1, 0)])
    ctx.set_pipeline(ov, sp)
    # (round 5) now and then the staged-lane sampler phase in front of the fused kernel: lone (2048) / paired (32768) chains
    form = rng.choice([0, 0, 0, 2048, 32768])
    ctx.set_debug_flags(form)
    c0 = torch.zeros((B, npr, n), dtype=torch.int32, device=dev); c1 = torch.zeros_like(c0)
    st = torch.zeros(B, dtype=torch.uint8, device=dev)
    if mode == "sym":
        ctx.encrypt_sym(T(vals), T(ss), T(sd), c0, c1, status=st)
        ok, e0, e1 = o.encrypt_sym_batch(vals, ss, sd, sk, nthreads=thr)
    else:
        ctx.encrypt_asym(T(vals), T(sd), c0, c1, status=st)
        ok, e0, e1 = o.encrypt_asym_batch(vals, sd, pk0, pk1, nthreads=thr)
    torch.cuda.synchronize()
    g0, g1 = c0.cpu().numpy().view(np.uint32), c1.cpu().numpy().view(np.uint32)
    if not (ok and bool(st.all()) and np.array_equal(g0, e0) and np.array_equal(g1, e1)):
        print(f"MISMATCH seed={master} case={cases} n={n} np={npr} B={B} mode={mode} pipe=({ov},{sp}) form={form} case_seed={seed}", flush=True)
        sys.exit(1)
    cases += 1; cts += B
    if cases % 5 == 0: print(f"{cases} cases, {cts} ciphertexts, {time.time()-t0:.0f}s", flush=True)
print(f"fuzz_mid ok: {cases} cases, {cts} ciphertexts bit-exact in {time.time()-t0:.0f}s (master seed {master})")
