#!/usr/bin/env python3
"""Randomised parity soak (GPU): random parameter sets, batch sizes, seeds, value distributions,
pipeline shapes and host-chunk sizes, every ciphertext compared bit for bit with the CPU oracle.
Runs for FUZZ_SECONDS (default 150).  Not part of the test suite; a failure prints the seed."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import vectors as V
import __graft_entry__ as ge
from oracle.pyoracle import Oracle
pkg = ge.load_package()
dev = torch.device("cuda:0")
budget = float(os.environ.get("FUZZ_SECONDS", "150"))
master = int(os.environ.get("FUZZ_SEED", str(int(time.time()))))
print("fuzz master seed", master, flush=True)
rng = random.Random(master)
SHAPES = [(1024, 1), (2048, 1), (4096, 1), (4096, 2), (4096, 3), (8192, 2), (8192, 6), (16384, 3), (16384, 6)]
t0 = time.time(); cases = cts = 0
ctxs = {}
def T(a): return torch.from_numpy(np.ascontiguousarray(a)).to(dev)
while time.time() - t0 < budget:
    n, npr = rng.choice(SHAPES)
    big = n >= 8192
    B = rng.choice([1, 2, 3, 63, 64, 65, 127, 128, 129, 200, 257, 300] if not big else [1, 2, 5, 33, 64, 65, 70])
    mode = rng.choice(["sym", "sym", "asym", "host", "hostasym", "stage"])
    case_seed = rng.getrandbits(32)
    nr = np.random.default_rng(case_seed)
    key = (n, npr)
    if key not in ctxs:
        ctx = pkg.Context(n, npr); sk = V.secret_key(n, seed=n + npr); ctx.set_secret_key(sk)
        o = Oracle(n, npr)
        pk0, pk1 = o.gen_pk(sk, bytes(range(64)), bytes(range(1, 65)))
        ctx.set_public_key(pk0, pk1)
        ctxs[key] = (ctx, o, sk, pk0, pk1)
    ctx, o, sk, pk0, pk1 = ctxs[key]
    kind = rng.randrange(6)
    if kind == 0:
        vals = V.bench_values(B, n, seed=case_seed)
    elif kind == 1:
        vals = (nr.standard_normal((B, n // 2)) * 3).astype(np.float32)
    elif kind == 2:
        vals = np.zeros((B, n // 2), dtype=np.float32); vals[:, nr.integers(0, n // 2)] = 1
    elif kind == 3:
        vals = nr.uniform(-30, 30, (B, n // 2)).astype(np.float32)
    elif kind == 4:   # rows of mixed magnitude: the fast kernels decline some plaintexts of the batch, not others
        vals = (nr.uniform(-30, 30, (B, n // 2)) * nr.choice([0.01, 1.0, 100.0], (B, 1))).astype(np.float32)
    else:   # NaN / Inf / FLT_MAX / subnormal / -0.0 values in some rows (accepted as INT64_MIN or rejected,
            # ckks_common.c:195; the general kernels' EXACT transform), ordinary rows between them
        vals = nr.uniform(-30, 30, (B, n // 2)).astype(np.float32)
        spec = np.array([np.inf, -np.inf, np.nan, 3.4028235e38, -3.4028235e38, 1e-45, -0.0, 1e-39], dtype=np.float32)
        for b in range(B):
            if nr.random() < 0.5:
                k = int(nr.integers(1, 6))
                pool = spec[nr.choice(len(spec), size=int(nr.integers(1, 4)), replace=False)]
                vals[b, nr.choice(n // 2, size=k, replace=False)] = nr.choice(pool, size=k)
                if nr.random() < 0.15:
                    vals[b, :] = nr.choice(pool, size=n // 2)
    ss = nr.integers(0, 256, (B, 64), dtype=np.uint8); sd = nr.integers(0, 256, (B, 64), dtype=np.uint8)
    ov, sp = rng.choice([(1, 2), (1, 2), (0, 0), (1, 0), (0, 1), (1, 1)])
    ctx.set_pipeline(ov, sp)
    ctx.set_reject_list_capacity(rng.choice([0, 0, 0, 3, 40]) or max(256, n // 16))
    # sampler forms: automatic / lane per ciphertext / staged (lane pairs + candidate kernel) / wave per ciphertext
    form = rng.choice([0, 0, 32, 512, 64])
    ctx.set_debug_flags(form)
    desc = f"seed={master} case={cases} n={n} np={npr} B={B} mode={mode} vals={kind} pipe=({ov},{sp}) form={form} case_seed={case_seed}"
    if mode == "stage":
        # stage-level operators: uniform sampler with random start counters, ternary -> CBD chain,
        # NTT -> INTT round trip against the oracle's forward NTT, PRNG blocks of random lengths
        import hashlib, struct
        Bs = min(B, 70)
        seeds = nr.integers(0, 256, (Bs, 64), dtype=np.uint8)
        cin = nr.integers(0, 2 ** 40, Bs, dtype=np.int64)
        out = torch.zeros((Bs, npr, n), dtype=torch.int32, device=dev)
        cout = torch.zeros(Bs, dtype=torch.int64, device=dev)
        ctx.sample_uniform(T(seeds), out, ctr_in=T(cin), ctr_out=cout)
        codes = torch.zeros((Bs, n), dtype=torch.int8, device=dev); tctr = torch.zeros(Bs, dtype=torch.int64, device=dev)
        ctx.sample_ternary(T(seeds), codes, tctr)
        err = torch.zeros((Bs, 2 * n), dtype=torch.int8, device=dev)
        ctx.sample_cbd(T(seeds), err, 2 * (n // 16), ctr_base=tctr)
        j = rng.randrange(npr); q = int(o.q[j])
        polys = nr.integers(0, q, (Bs, n), dtype=np.uint64).astype(np.uint32)
        tp = T(polys.view(np.int32)).clone(); ctx.ntt(j, tp); fwd = tp.clone(); ctx.intt(j, tp)
        outlen = rng.choice([1, 3, 8, 96, 135, 136, 137, 272, 1000])
        pctr = nr.integers(0, 2 ** 62, Bs, dtype=np.int64)
        pout = torch.zeros((Bs, outlen), dtype=torch.uint8, device=dev)
        ctx.prng_blocks(T(seeds), T(pctr), pout, outlen)
        torch.cuda.synchronize()
        g, gc = out.cpu().numpy().view(np.uint32), cout.cpu().numpy()
        gcodes, gt, ge = codes.cpu().numpy(), tctr.cpu().numpy(), err.cpu().numpy()
        gf, gi, gp = fwd.cpu().numpy().view(np.uint32), tp.cpu().numpy().view(np.uint32), pout.cpu().numpy()
        for b in sorted(set([0, Bs - 1, rng.randrange(Bs)])):
            c = int(cin[b])
            for jj in range(npr):
                a, c = o.sample_uniform(jj, seeds[b].tobytes(), c)
                if not (g[b, jj] == a).all(): print("MISMATCH uniform", desc, b, jj); sys.exit(1)
            if int(gc[b]) != c: print("MISMATCH uniform ctr", desc, b); sys.exit(1)
            u, c = o.sample_ternary_small(seeds[b].tobytes(), 0)
            e0, c2 = o.cbd_int8(seeds[b].tobytes(), c); e1, _ = o.cbd_int8(seeds[b].tobytes(), c2)
            if not ((ctx.pack_ternary(gcodes[b]) == u).all() and int(gt[b]) == c and (ge[b, :n] == e0).all() and (ge[b, n:] == e1).all()):
                print("MISMATCH ternary/cbd", desc, b); sys.exit(1)
            if not ((gf[b] == o.ntt(polys[b], j)).all() and (gi[b] == polys[b]).all()):
                print("MISMATCH ntt/intt", desc, b); sys.exit(1)
            if gp[b].tobytes() != hashlib.shake_256(seeds[b].tobytes() + struct.pack("<Q", int(pctr[b]))).digest(outlen):
                print("MISMATCH prng", desc, b); sys.exit(1)
            cts += 1
        cases += 1
        continue
    if mode in ("sym", "asym"):
        c0 = torch.zeros((B, npr, n), dtype=torch.int32, device=dev); c1 = torch.zeros_like(c0)
        st = torch.zeros(B, dtype=torch.uint8, device=dev)
        if mode == "sym": ctx.encrypt_sym(T(vals), T(ss), T(sd), c0, c1, status=st)
        else: ctx.encrypt_asym(T(vals), T(sd), c0, c1, status=st)
        torch.cuda.synchronize()
        g0, g1, gs = c0.cpu().numpy().view(np.uint32), c1.cpu().numpy().view(np.uint32), st.cpu().numpy()
    else:
        ctx.set_host_chunk(rng.choice([0, 0, 1, 7, 64, 100]))
        r = ctx.encrypt_sym_host(vals, ss, sd) if mode == "host" else ctx.encrypt_asym_host(vals, sd)
        g0, g1, gs = r["c0"], r["c1"], r["status"]
        ctx.set_host_chunk(0)
    check = range(B) if B <= 70 else sorted(set([0, 1, 62, 63, 64, 65, B - 2, B - 1] + [rng.randrange(B) for _ in range(24)]))
    for b in check:
        e = o.encrypt_sym(vals[b], ss[b].tobytes(), sd[b].tobytes(), sk) if mode in ("sym", "host") else \
            o.encrypt_asym(vals[b], sd[b].tobytes(), pk0, pk1)
        if not (e["ok"] == bool(gs[b]) and (not e["ok"] or ((g0[b] == e["c0"]).all() and (g1[b] == e["c1"]).all()))):
            print("MISMATCH", desc, "ct", b, flush=True); sys.exit(1)
        cts += 1
    cases += 1
    if cases % 20 == 0: print(f"{cases} cases, {cts} ciphertexts checked, {time.time()-t0:.0f}s", flush=True)
print(f"fuzz ok: {cases} cases, {cts} ciphertexts bit-exact in {time.time()-t0:.0f}s (master seed {master})")
