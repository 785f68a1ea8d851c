#!/usr/bin/env python3
"""Per-call latency of the symmetric LOWER surface (ckks_encode_base, ckks_sym_init, ckks_encode_encrypt_sym per
prime) as a caller of the reference's prototypes sees it -- the sequence of device/bench/bench_sym.c:96-130, timed
call by call.  usage (GPU box): python tools/lower_sym_latency.py [n=4096] [nprimes=3] [reps=12]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import __graft_entry__ as ge
import vectors as V
from test_gpu_lower import Parms, Prng, _vp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
npr = int(sys.argv[2]) if len(sys.argv) > 2 else 3
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
pkg = ge.load_package()
L = pkg.lib()
L.next_modulus.restype = C.c_bool
L.ckks_encode_base.restype = C.c_bool
P = Parms()
imap = np.zeros(n, np.uint16)
L.ckks_setup(C.c_size_t(n), C.c_size_t(npr), _vp(imap), C.byref(P))
sk = V.secret_key(n)
rng = np.random.default_rng(1)
c0, c1, ntt_pte, roots = (np.zeros(n, np.uint32) for _ in range(4))
conj = np.zeros(2 * n, np.float64)          # se_complex[n]; the int64 plaintext aliases its first half
for rep in range(reps):
    v = rng.uniform(-10, 10, n // 2).astype(np.float32)
    t = [time.perf_counter()]
    L.ckks_encode_base(C.byref(P), _vp(v), C.c_size_t(n // 2), _vp(imap), None, _vp(conj))
    t.append(time.perf_counter())
    pa, pe = Prng(), Prng()
    L.ckks_sym_init(C.byref(P), None, None, C.byref(pa), C.byref(pe), _vp(conj))
    t.append(time.perf_counter())
    for j in range(npr):
        L.ckks_encode_encrypt_sym(C.byref(P), _vp(conj), None, C.byref(pa), _vp(sk), _vp(ntt_pte), _vp(roots), _vp(c0),
                                  _vp(c1), None, None)
        t.append(time.perf_counter())
        L.next_modulus(C.byref(P))
    d = [(b - a) * 1e6 for a, b in zip(t, t[1:])]
    print("rep %2d  encode %6.0f  init %6.0f  primes %s  total %7.0f us   end ctr %d" %
          (rep, d[0], d[1], " ".join("%6.0f" % x for x in d[2:]), sum(d), pa.counter), flush=True)
