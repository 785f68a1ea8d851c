"""TEST HARNESS ONLY: a stand-in for seal_embedded_amd.Context whose records come from the CPU oracle,
loaded by bench.py when SE_BENCH_STUB=stub_context is set (tests/test_sharding.py).  It lets the
rank logic of bench.py -- `first = rank * B` sharding, barrier-bracketed timing with the max over
ranks, the point-to-point gather and the JSON line for n_gpus = 2 -- run under torchrun with gloo on
a box without GPUs.  Never part of the product path: the library has no CPU implementation."""
import numpy as np

from oracle import pyoracle


class Context:
    def __init__(self, n, nprimes, device=0):
        self.o = pyoracle.Oracle(n, nprimes)
        self.n, self.np = n, nprimes
        self.sk = None
        self.pk = None
        self.calls = 0

    def set_secret_key(self, sk):
        self.sk = np.ascontiguousarray(sk, dtype=np.uint8)

    def gen_public_key(self, sk, pk_seed, ep_seed):
        return self.o.gen_pk(sk, pk_seed, ep_seed)

    def set_public_key(self, pk0, pk1):
        self.pk = (pk0, pk1)

    def reserve(self, B):
        pass

    def encrypt_sym(self, values, share_seeds, seeds, c0, c1, ntt_pte=None, pte=None, status=None):
        ok, e0, e1 = self.o.encrypt_sym_batch(values.numpy(), share_seeds.numpy(), seeds.numpy(), self.sk)
        c0.numpy().view(np.uint32)[...] = e0
        c1.numpy().view(np.uint32)[...] = e1
        if status is not None:
            status.fill_(1 if ok else 0)
        self.calls += 1

    def encrypt_asym(self, values, seeds, c0, c1, ntt_pte=None, pte=None, status=None):
        ok, e0, e1 = self.o.encrypt_asym_batch(values.numpy(), seeds.numpy(), *self.pk)
        c0.numpy().view(np.uint32)[...] = e0
        c1.numpy().view(np.uint32)[...] = e1
        if status is not None:
            status.fill_(1 if ok else 0)
        self.calls += 1

    def encode_ntt(self, values, out, pte=None, status=None):
        ok, e = self.o.encode_ntt_batch(values.numpy())
        out.numpy().view(np.uint32)[...] = e
        if status is not None:
            status.fill_(1 if ok else 0)
        self.calls += 1

    def set_debug_flags(self, flags):
        pass

    def set_profiling(self, on=True):
        pass

    def stage_ms(self, reset=True):
        return {}

    def close(self):
        pass
