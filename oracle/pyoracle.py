"""ctypes bindings for the CPU oracle (libse_oracle.so) and, when present, the compiled
reference (oracle/_ref/libse_ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product (seal-embedded_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libse_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libse_ref.so")
REFERENCE_DIR = os.environ.get("SE_REFERENCE_DIR", "/root/reference")

MAX_PRIMES = 13

u8p = C.POINTER(C.c_uint8)
u16p = C.POINTER(C.c_uint16)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
i8p = C.POINTER(C.c_int8)
i64p = C.POINTER(C.c_int64)
f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


def build(ref=True):
    """Compile the oracle (and the reference harness when /root/reference is present)."""
    subprocess.check_call(["make", "-s", "-C", HERE])
    if ref and os.path.isdir(os.path.join(REFERENCE_DIR, "device", "lib")):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref", "SE_REFERENCE_DIR=" + REFERENCE_DIR])
        # the reference's own test functions relinked against the product library (tests/test_gpu_reftests.py)
        lib = os.path.join(os.path.dirname(HERE), "seal-embedded_amd", "lib", "libseal_embedded_amd.so")
        if os.path.exists(lib):
            subprocess.check_call(["make", "-s", "-C", HERE, "reftests", "refbench",
                                   "SE_REFERENCE_DIR=" + REFERENCE_DIR])


class SeoParams(C.Structure):
    _fields_ = [("n", C.c_size_t), ("logn", C.c_size_t), ("nprimes", C.c_size_t),
                ("q", C.c_uint32 * MAX_PRIMES), ("cr_lo", C.c_uint32 * MAX_PRIMES),
                ("cr_hi", C.c_uint32 * MAX_PRIMES), ("psi", C.c_uint32 * MAX_PRIMES),
                ("scale", C.c_double)]


def _seed(s):
    a = np.ascontiguousarray(np.frombuffer(bytes(s), dtype=np.uint8))
    assert a.size == 64
    return a


class Oracle:
    """Our C restatement, one parameter set."""

    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            if not os.path.exists(ORACLE_SO):
                build(ref=False)
            L = C.CDLL(ORACLE_SO)
            L.seo_params_init.restype = C.c_int
            L.seo_params_init.argtypes = [C.POINTER(SeoParams), C.c_size_t, C.c_size_t]
            for name in ("seo_barrett32",):
                getattr(L, name).restype = C.c_uint32
                getattr(L, name).argtypes = [C.c_uint32, C.POINTER(SeoParams), C.c_size_t]
            L.seo_barrett64.restype = C.c_uint32
            L.seo_barrett64.argtypes = [C.c_uint64, C.POINTER(SeoParams), C.c_size_t]
            L.seo_mul_mod.restype = C.c_uint32
            L.seo_mul_mod.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(SeoParams), C.c_size_t]
            for name in ("seo_add_mod", "seo_sub_mod"):
                getattr(L, name).restype = C.c_uint32
                getattr(L, name).argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
            L.seo_neg_mod.restype = C.c_uint32
            L.seo_neg_mod.argtypes = [C.c_uint32, C.c_uint32]
            L.seo_shake256.argtypes = [u8p, C.c_size_t, u8p, C.c_size_t]
            L.seo_prng_block.argtypes = [u8p, C.c_uint64, u8p, C.c_size_t]
            L.seo_keccak_f1600.argtypes = [u64p]
            L.seo_index_map.argtypes = [C.c_size_t, C.c_size_t, u16p]
            L.seo_ifft_twiddles.argtypes = [C.c_size_t, C.c_size_t, f64p]
            L.seo_ifft_inpl.argtypes = [f64p, C.c_size_t, C.c_size_t]
            L.seo_encode_ex.restype = C.c_size_t
            L.seo_encode_ex.argtypes = [C.POINTER(SeoParams), f32p, C.c_size_t, u16p, i64p]
            L.seo_encode.restype = C.c_int
            L.seo_encode.argtypes = [C.POINTER(SeoParams), f32p, C.c_size_t, u16p, i64p]
            L.seo_cbd_add.argtypes = [i64p, C.c_size_t, u8p, u64p]
            L.seo_cbd_int8.argtypes = [i8p, C.c_size_t, u8p, u64p]
            L.seo_sample_uniform.argtypes = [C.POINTER(SeoParams), C.c_size_t, u8p, u64p, u32p]
            L.seo_sample_ternary_small.argtypes = [C.c_size_t, u8p, u64p, u8p]
            L.seo_expand_ternary.argtypes = [u8p, C.c_size_t, C.c_uint32, u32p]
            L.seo_ntt_roots.argtypes = [C.POINTER(SeoParams), C.c_size_t, u32p]
            L.seo_ntt_inpl.argtypes = [C.POINTER(SeoParams), C.c_size_t, u32p, u32p]
            L.seo_reduce_pte.argtypes = [C.POINTER(SeoParams), C.c_size_t, i64p, u32p]
            L.seo_reduce_e_small.argtypes = [C.POINTER(SeoParams), C.c_size_t, i8p, u32p]
            L.seo_intt_inpl.argtypes = [C.POINTER(SeoParams), C.c_size_t, u32p]
            L.seo_fft_inpl.argtypes = [f64p, C.c_size_t, C.c_size_t]
            L.seo_decrypt.argtypes = [C.POINTER(SeoParams), C.c_size_t, u32p, u32p, u32p, u32p]
            L.seo_decode.argtypes = [C.POINTER(SeoParams), C.c_size_t, u16p, u32p, C.c_size_t, f32p]
            L.seo_encrypt_sym.restype = C.c_int
            L.seo_encrypt_sym.argtypes = [C.POINTER(SeoParams), u16p, f32p, C.c_size_t, u8p, u8p,
                                          u8p, u32p, u32p, i64p, u32p, u64p]
            L.seo_encrypt_asym.restype = C.c_int
            L.seo_encrypt_asym.argtypes = [C.POINTER(SeoParams), u16p, f32p, C.c_size_t, u8p,
                                           u32p, u32p, u32p, u32p, i64p, u8p, i8p, u64p]
            L.seo_gen_pk.argtypes = [C.POINTER(SeoParams), u8p, u8p, u8p, u32p, u32p]
            L.seo_encrypt_sym_batch.restype = C.c_int
            L.seo_encrypt_sym_batch.argtypes = [C.POINTER(SeoParams), f32p, C.c_size_t, u8p, u8p,
                                                u8p, u32p, u32p, C.c_int]
            L.seo_encrypt_asym_batch.restype = C.c_int
            L.seo_encrypt_asym_batch.argtypes = [C.POINTER(SeoParams), f32p, C.c_size_t, u8p, u32p, u32p,
                                                 u32p, u32p, C.c_int]
            L.seo_encode_ntt_batch.restype = C.c_int
            L.seo_encode_ntt_batch.argtypes = [C.POINTER(SeoParams), f32p, C.c_size_t, u32p, C.c_int]
            L.seo_fnv1a64.restype = C.c_uint64
            L.seo_fnv1a64.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
            cls._lib = L
        return cls._lib

    def __init__(self, n, nprimes):
        self.L = self.lib()
        self.p = SeoParams()
        rc = self.L.seo_params_init(C.byref(self.p), n, nprimes)
        if rc != 0:
            raise ValueError(f"unsupported parameter set n={n} nprimes={nprimes} (rc={rc})")
        self.n, self.np, self.logn = n, nprimes, int(self.p.logn)
        self.q = [int(self.p.q[j]) for j in range(nprimes)]
        self.scale = float(self.p.scale)
        self.map = np.zeros(n, dtype=np.uint16)
        self.L.seo_index_map(n, self.logn, _p(self.map, u16p))

    # -- word level
    def barrett32(self, x, j=0):
        return int(self.L.seo_barrett32(x, C.byref(self.p), j))

    def barrett64(self, x, j=0):
        return int(self.L.seo_barrett64(x, C.byref(self.p), j))

    def mul_mod(self, a, b, j=0):
        return int(self.L.seo_mul_mod(a, b, C.byref(self.p), j))

    def add_mod(self, a, b, j=0):
        return int(self.L.seo_add_mod(a, b, self.q[j]))

    def neg_mod(self, a, j=0):
        return int(self.L.seo_neg_mod(a, self.q[j]))

    # -- PRNG
    @classmethod
    def shake256(cls, data, outlen):
        L = cls.lib()
        inp = np.frombuffer(bytes(data), dtype=np.uint8).copy()
        out = np.zeros(outlen, dtype=np.uint8)
        L.seo_shake256(_p(out, u8p), outlen, _p(inp, u8p), inp.size)
        return out.tobytes()

    @classmethod
    def prng_block(cls, seed, ctr, outlen):
        L = cls.lib()
        out = np.zeros(outlen, dtype=np.uint8)
        L.seo_prng_block(_p(_seed(seed), u8p), ctr, _p(out, u8p), outlen)
        return out.tobytes()

    @classmethod
    def keccak_f1600(cls, state):
        st = np.array(state, dtype=np.uint64).copy()
        cls.lib().seo_keccak_f1600(_p(st, u64p))
        return st

    # -- encode
    def twiddles(self):
        w = np.zeros(2 * self.n, dtype=np.float64)
        self.L.seo_ifft_twiddles(self.n, self.logn, _p(w, f64p))
        return w

    def ifft(self, x_complex):
        x = np.ascontiguousarray(np.asarray(x_complex, dtype=np.complex128)).view(np.float64).copy()
        self.L.seo_ifft_inpl(_p(x, f64p), self.n, self.logn)
        return x.view(np.complex128)

    def encode(self, values):
        v = np.zeros(self.n // 2, dtype=np.float32)
        vv = np.asarray(values, dtype=np.float32).ravel()
        v[:vv.size] = vv
        out = np.zeros(self.n, dtype=np.int64)
        ok = self.L.seo_encode(C.byref(self.p), _p(v, f32p), self.n // 2, _p(self.map, u16p),
                               _p(out, i64p))
        return bool(ok), out

    def encode_ex(self, values):
        """(index of the first coefficient that fails the overflow test or n, coefficients): the first
        `index` entries are what the reference's in-place loop has converted when it returns false."""
        v = np.zeros(self.n // 2, dtype=np.float32)
        vv = np.asarray(values, dtype=np.float32).ravel()
        v[:vv.size] = vv
        out = np.zeros(self.n, dtype=np.int64)
        idx = self.L.seo_encode_ex(C.byref(self.p), _p(v, f32p), self.n // 2, _p(self.map, u16p),
                                   _p(out, i64p))
        return int(idx), out

    # -- samplers (return (array, next_ctr))
    def cbd_int8(self, seed, ctr=0):
        out = np.zeros(self.n, dtype=np.int8)
        c = C.c_uint64(ctr)
        self.L.seo_cbd_int8(_p(out, i8p), self.n, _p(_seed(seed), u8p), C.byref(c))
        return out, int(c.value)

    def cbd_add(self, poly, seed, ctr=0):
        out = np.array(poly, dtype=np.int64).copy()
        c = C.c_uint64(ctr)
        self.L.seo_cbd_add(_p(out, i64p), self.n, _p(_seed(seed), u8p), C.byref(c))
        return out, int(c.value)

    def sample_uniform(self, j, seed, ctr=0):
        out = np.zeros(self.n, dtype=np.uint32)
        c = C.c_uint64(ctr)
        self.L.seo_sample_uniform(C.byref(self.p), j, _p(_seed(seed), u8p), C.byref(c),
                                  _p(out, u32p))
        return out, int(c.value)

    def sample_ternary_small(self, seed, ctr=0):
        out = np.zeros(self.n // 4, dtype=np.uint8)
        c = C.c_uint64(ctr)
        self.L.seo_sample_ternary_small(self.n, _p(_seed(seed), u8p), C.byref(c), _p(out, u8p))
        return out, int(c.value)

    def expand_ternary(self, packed, j):
        pk = np.ascontiguousarray(packed, dtype=np.uint8)
        out = np.zeros(self.n, dtype=np.uint32)
        self.L.seo_expand_ternary(_p(pk, u8p), self.n, self.q[j], _p(out, u32p))
        return out

    # -- NTT
    def ntt_roots(self, j):
        r = np.zeros(self.n, dtype=np.uint32)
        self.L.seo_ntt_roots(C.byref(self.p), j, _p(r, u32p))
        return r

    def ntt(self, vec, j):
        v = np.array(vec, dtype=np.uint32).copy()
        r = self.ntt_roots(j)
        self.L.seo_ntt_inpl(C.byref(self.p), j, _p(r, u32p), _p(v, u32p))
        return v

    def reduce_pte(self, x, j):
        xin = np.ascontiguousarray(x, dtype=np.int64)
        out = np.zeros(self.n, dtype=np.uint32)
        self.L.seo_reduce_pte(C.byref(self.p), j, _p(xin, i64p), _p(out, u32p))
        return out

    def reduce_e_small(self, e, j):
        ein = np.ascontiguousarray(e, dtype=np.int8)
        out = np.zeros(self.n, dtype=np.uint32)
        self.L.seo_reduce_e_small(C.byref(self.p), j, _p(ein, i8p), _p(out, u32p))
        return out

    # -- verification side
    def intt(self, vec, j):
        v = np.array(vec, dtype=np.uint32).copy()
        self.L.seo_intt_inpl(C.byref(self.p), j, _p(v, u32p))
        return v

    def fft(self, x_complex):
        x = np.ascontiguousarray(np.asarray(x_complex, dtype=np.complex128)).view(np.float64).copy()
        self.L.seo_fft_inpl(_p(x, f64p), self.n, self.logn)
        return x.view(np.complex128)

    def decrypt(self, c0, c1, ntt_s, j):
        a = np.ascontiguousarray(c0, dtype=np.uint32)
        b = np.ascontiguousarray(c1, dtype=np.uint32)
        s = np.ascontiguousarray(ntt_s, dtype=np.uint32)
        out = np.zeros(self.n, dtype=np.uint32)
        self.L.seo_decrypt(C.byref(self.p), j, _p(a, u32p), _p(b, u32p), _p(s, u32p), _p(out, u32p))
        return out

    def decode(self, pt, j, values_len=None):
        values_len = self.n // 2 if values_len is None else values_len
        x = np.ascontiguousarray(pt, dtype=np.uint32)
        out = np.zeros(values_len, dtype=np.float32)
        self.L.seo_decode(C.byref(self.p), j, _p(self.map, u16p), _p(x, u32p), values_len,
                          _p(out, f32p))
        return out

    # -- whole path
    def encrypt_sym(self, values, share_seed, seed, sk_packed):
        v = np.ascontiguousarray(values, dtype=np.float32).ravel()
        sk = np.ascontiguousarray(sk_packed, dtype=np.uint8)
        c0 = np.zeros((self.np, self.n), dtype=np.uint32)
        c1 = np.zeros((self.np, self.n), dtype=np.uint32)
        pte = np.zeros(self.n, dtype=np.int64)
        ntt_pte = np.zeros((self.np, self.n), dtype=np.uint32)
        ctr = C.c_uint64(0)
        ok = self.L.seo_encrypt_sym(C.byref(self.p), _p(self.map, u16p), _p(v, f32p), v.size,
                                    _p(_seed(share_seed), u8p), _p(_seed(seed), u8p),
                                    _p(sk, u8p), _p(c0, u32p), _p(c1, u32p), _p(pte, i64p),
                                    _p(ntt_pte, u32p), C.byref(ctr))
        return dict(ok=bool(ok), c0=c0, c1=c1, pte=pte, ntt_pte=ntt_pte, end_ctr=int(ctr.value))

    def encrypt_asym(self, values, seed, pk0, pk1):
        v = np.ascontiguousarray(values, dtype=np.float32).ravel()
        pk0 = np.ascontiguousarray(pk0, dtype=np.uint32)
        pk1 = np.ascontiguousarray(pk1, dtype=np.uint32)
        c0 = np.zeros((self.np, self.n), dtype=np.uint32)
        c1 = np.zeros((self.np, self.n), dtype=np.uint32)
        pte = np.zeros(self.n, dtype=np.int64)
        u = np.zeros(self.n // 4, dtype=np.uint8)
        e1 = np.zeros(self.n, dtype=np.int8)
        ctr = C.c_uint64(0)
        ok = self.L.seo_encrypt_asym(C.byref(self.p), _p(self.map, u16p), _p(v, f32p), v.size,
                                     _p(_seed(seed), u8p), _p(pk0, u32p), _p(pk1, u32p),
                                     _p(c0, u32p), _p(c1, u32p), _p(pte, i64p), _p(u, u8p),
                                     _p(e1, i8p), C.byref(ctr))
        return dict(ok=bool(ok), c0=c0, c1=c1, pte=pte, u=u, e1=e1, end_ctr=int(ctr.value))

    def gen_pk(self, sk_packed, pk_seed, ep_seed):
        sk = np.ascontiguousarray(sk_packed, dtype=np.uint8)
        pk0 = np.zeros((self.np, self.n), dtype=np.uint32)
        pk1 = np.zeros((self.np, self.n), dtype=np.uint32)
        self.L.seo_gen_pk(C.byref(self.p), _p(sk, u8p), _p(_seed(pk_seed), u8p),
                          _p(_seed(ep_seed), u8p), _p(pk0, u32p), _p(pk1, u32p))
        return pk0, pk1

    def encrypt_sym_batch(self, values, share_seeds, seeds, sk_packed, nthreads=1, keep=True):
        v = np.ascontiguousarray(values, dtype=np.float32)
        B = v.shape[0]
        ss = np.ascontiguousarray(share_seeds, dtype=np.uint8)
        sd = np.ascontiguousarray(seeds, dtype=np.uint8)
        sk = np.ascontiguousarray(sk_packed, dtype=np.uint8)
        c0 = np.zeros((B, self.np, self.n), dtype=np.uint32) if keep else None
        c1 = np.zeros((B, self.np, self.n), dtype=np.uint32) if keep else None
        ok = self.L.seo_encrypt_sym_batch(C.byref(self.p), _p(v, f32p), B, _p(ss, u8p),
                                          _p(sd, u8p), _p(sk, u8p), _p(c0, u32p), _p(c1, u32p),
                                          nthreads)
        return bool(ok), c0, c1


    def encrypt_asym_batch(self, values, seeds, pk0, pk1, nthreads=1, keep=True):
        v = np.ascontiguousarray(values, dtype=np.float32)
        B = v.shape[0]
        sd = np.ascontiguousarray(seeds, dtype=np.uint8)
        pk0 = np.ascontiguousarray(pk0, dtype=np.uint32)
        pk1 = np.ascontiguousarray(pk1, dtype=np.uint32)
        c0 = np.zeros((B, self.np, self.n), dtype=np.uint32) if keep else None
        c1 = np.zeros((B, self.np, self.n), dtype=np.uint32) if keep else None
        ok = self.L.seo_encrypt_asym_batch(C.byref(self.p), _p(v, f32p), B, _p(sd, u8p), _p(pk0, u32p),
                                           _p(pk1, u32p), _p(c0, u32p), _p(c1, u32p), nthreads)
        return bool(ok), c0, c1

    def encode_ntt_batch(self, values, nthreads=1, keep=True):
        v = np.ascontiguousarray(values, dtype=np.float32)
        B = v.shape[0]
        out = np.zeros((B, self.np, self.n), dtype=np.uint32) if keep else None
        ok = self.L.seo_encode_ntt_batch(C.byref(self.p), _p(v, f32p), B, _p(out, u32p), nthreads)
        return bool(ok), out


def host_threads():
    """Host threads this process may really use: affinity mask clipped by the cgroup CPU quota."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, -(-int(quota) // int(period))))
    except Exception:
        pass
    return cores


def host_threads_why():
    """Why host_threads() is what it is, for the bench line: affinity mask, cgroup quota, hardware threads."""
    try:
        aff = len(os.sched_getaffinity(0))
    except AttributeError:
        aff = os.cpu_count() or 1
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = int(q) / int(period)
    except Exception:
        pass
    hw = os.cpu_count() or aff
    t = host_threads()
    if quota is not None and t < aff:
        return f"{t} = cgroup CPU quota ({quota:g} CPUs) of this process on a box with {hw} hardware threads"
    if aff < hw:
        return f"{t} = affinity mask of this process ({aff} of {hw} hardware threads)"
    return f"{t} = all {hw} hardware threads of the box"


def fnv1a64(data, h=0):
    L = Oracle.lib()
    b = np.frombuffer(bytes(data), dtype=np.uint8)
    return int(L.seo_fnv1a64(b.ctypes.data_as(C.c_void_p), b.size, h))


def ref_available():
    return os.path.exists(REF_SO)


class Reference:
    """The compiled reference (oracle/_ref/libse_ref.so), one instance per parameter set."""

    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            L = C.CDLL(REF_SO)
            L.refh_open.restype = C.c_void_p
            L.refh_open.argtypes = [C.c_size_t, C.c_size_t, C.c_int]
            L.refh_close.argtypes = [C.c_void_p]
            L.refh_scale.restype = C.c_double
            L.refh_scale.argtypes = [C.c_void_p]
            L.refh_moduli.argtypes = [C.c_void_p, u32p, u32p, u32p]
            L.refh_index_map.argtypes = [C.c_void_p, u16p]
            L.refh_set_sk.argtypes = [C.c_void_p, u8p]
            L.refh_encode_ex.restype = C.c_long
            L.refh_encode_ex.argtypes = [C.c_void_p, f32p, C.c_size_t, i64p]
            L.refh_encode.restype = C.c_int
            L.refh_encode.argtypes = [C.c_void_p, f32p, C.c_size_t, i64p]
            L.refh_shake256.argtypes = [u8p, C.c_size_t, u8p, C.c_size_t]
            L.refh_prng_block.argtypes = [u8p, C.c_uint64, u8p, C.c_size_t]
            for nm, last in (("refh_sample_uniform", u32p),):
                getattr(L, nm).restype = C.c_uint64
                getattr(L, nm).argtypes = [C.c_void_p, C.c_size_t, u8p, C.c_uint64, last]
            L.refh_sample_ternary_small.restype = C.c_uint64
            L.refh_sample_ternary_small.argtypes = [C.c_void_p, u8p, C.c_uint64, u8p]
            L.refh_cbd_int8.restype = C.c_uint64
            L.refh_cbd_int8.argtypes = [C.c_void_p, u8p, C.c_uint64, i8p]
            L.refh_cbd_add.restype = C.c_uint64
            L.refh_cbd_add.argtypes = [C.c_void_p, u8p, C.c_uint64, i64p]
            L.refh_expand_ternary.argtypes = [C.c_void_p, C.c_size_t, u8p, u32p]
            L.refh_ntt_roots.argtypes = [C.c_void_p, C.c_size_t, u32p]
            L.refh_ntt.argtypes = [C.c_void_p, C.c_size_t, u32p]
            L.refh_reduce_pte.argtypes = [C.c_void_p, C.c_size_t, i64p, u32p]
            L.refh_reduce_e_small.argtypes = [C.c_void_p, C.c_size_t, i8p, u32p]
            L.refh_barrett32.restype = C.c_uint32
            L.refh_barrett32.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32]
            L.refh_barrett64.restype = C.c_uint32
            L.refh_barrett64.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
            L.refh_mul_mod.restype = C.c_uint32
            L.refh_mul_mod.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32]
            L.refh_encrypt_sym.restype = C.c_int
            L.refh_encrypt_sym.argtypes = [C.c_void_p, f32p, C.c_size_t, u8p, u8p, u32p, u32p,
                                           u32p, i64p, u32p, u64p]
            L.refh_encrypt_asym.restype = C.c_int
            L.refh_encrypt_asym.argtypes = [C.c_void_p, f32p, C.c_size_t, u8p, u32p, u32p, u32p,
                                            u32p, i64p, u8p, i8p, u64p]
            L.refh_gen_pk.argtypes = [C.c_size_t, C.c_size_t, u8p, u8p, u8p, u32p, u32p]
            L.refh_api_encrypt.restype = C.c_long
            L.refh_api_encrypt.argtypes = [C.c_size_t, C.c_size_t, C.c_int, f32p, C.c_size_t, u8p,
                                           u8p, u8p, C.c_size_t, C.POINTER(C.c_size_t)]
            L.refh_intt.argtypes = [C.c_void_p, C.c_size_t, u32p]
            L.refh_decrypt.argtypes = [C.c_void_p, C.c_size_t, u32p, u32p, u32p, u32p]
            L.refh_decode.argtypes = [C.c_void_p, C.c_size_t, u32p, C.c_size_t, f32p]
            L.refh_print_to_file.argtypes = [C.c_char_p, C.c_char_p, u32p, C.c_size_t, f32p, C.c_size_t]
            L.refh_api_encrypt_print.restype = C.c_long
            L.refh_api_encrypt_print.argtypes = [C.c_size_t, C.c_size_t, C.c_int, f32p, C.c_size_t, u8p, u8p,
                                                 C.c_char_p]
            L.refh_encrypt_asym_batch.restype = C.c_int
            L.refh_encrypt_asym_batch.argtypes = [C.c_size_t, C.c_size_t, f32p, C.c_size_t, u8p, u32p,
                                                  u32p, u32p, u32p, C.c_int]
            L.refh_encode_ntt_batch.restype = C.c_int
            L.refh_encode_ntt_batch.argtypes = [C.c_size_t, C.c_size_t, f32p, C.c_size_t, u32p, C.c_int]
            L.refh_encrypt_sym_batch.restype = C.c_int
            L.refh_encrypt_sym_batch.argtypes = [C.c_size_t, C.c_size_t, f32p, C.c_size_t, u8p,
                                                 u8p, u8p, u32p, u32p, C.c_int]
            cls._lib = L
        return cls._lib

    def __init__(self, n, nprimes, asym=False):
        self.L = self.lib()
        self.n, self.np, self.asym = n, nprimes, asym
        self.h = C.c_void_p(self.L.refh_open(n, nprimes, 1 if asym else 0))

    def close(self):
        if self.h:
            self.L.refh_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def scale(self):
        return float(self.L.refh_scale(self.h))

    def moduli(self):
        q = np.zeros(self.np, dtype=np.uint32)
        lo = np.zeros(self.np, dtype=np.uint32)
        hi = np.zeros(self.np, dtype=np.uint32)
        self.L.refh_moduli(self.h, _p(q, u32p), _p(lo, u32p), _p(hi, u32p))
        return q, lo, hi

    def index_map(self):
        m = np.zeros(self.n, dtype=np.uint16)
        self.L.refh_index_map(self.h, _p(m, u16p))
        return m

    def set_sk(self, sk_packed):
        sk = np.ascontiguousarray(sk_packed, dtype=np.uint8)
        assert sk.size == self.n // 4
        self.L.refh_set_sk(self.h, _p(sk, u8p))

    def encode(self, values):
        v = np.ascontiguousarray(values, dtype=np.float32).ravel()
        out = np.zeros(self.n, dtype=np.int64)
        ok = self.L.refh_encode(self.h, _p(v, f32p), v.size, _p(out, i64p))
        return bool(ok), out

    def encode_ex(self, values):
        """(index at which ckks_encode_base returned false or n, the in-place buffer as int64)"""
        v = np.ascontiguousarray(values, dtype=np.float32).ravel()
        out = np.zeros(self.n, dtype=np.int64)
        idx = self.L.refh_encode_ex(self.h, _p(v, f32p), v.size, _p(out, i64p))
        return int(idx), out

    @classmethod
    def shake256(cls, data, outlen):
        L = cls.lib()
        inp = np.frombuffer(bytes(data), dtype=np.uint8).copy()
        out = np.zeros(outlen, dtype=np.uint8)
        L.refh_shake256(_p(out, u8p), outlen, _p(inp, u8p), inp.size)
        return out.tobytes()

    @classmethod
    def prng_block(cls, seed, ctr, outlen):
        L = cls.lib()
        out = np.zeros(outlen, dtype=np.uint8)
        L.refh_prng_block(_p(_seed(seed), u8p), ctr, _p(out, u8p), outlen)
        return out.tobytes()

    def sample_uniform(self, j, seed, ctr=0):
        out = np.zeros(self.n, dtype=np.uint32)
        c = self.L.refh_sample_uniform(self.h, j, _p(_seed(seed), u8p), ctr, _p(out, u32p))
        return out, int(c)

    def sample_ternary_small(self, seed, ctr=0):
        out = np.zeros(self.n // 4, dtype=np.uint8)
        c = self.L.refh_sample_ternary_small(self.h, _p(_seed(seed), u8p), ctr, _p(out, u8p))
        return out, int(c)

    def cbd_int8(self, seed, ctr=0):
        out = np.zeros(self.n, dtype=np.int8)
        c = self.L.refh_cbd_int8(self.h, _p(_seed(seed), u8p), ctr, _p(out, i8p))
        return out, int(c)

    def cbd_add(self, poly, seed, ctr=0):
        out = np.array(poly, dtype=np.int64).copy()
        c = self.L.refh_cbd_add(self.h, _p(_seed(seed), u8p), ctr, _p(out, i64p))
        return out, int(c)

    def expand_ternary(self, packed, j):
        pk = np.ascontiguousarray(packed, dtype=np.uint8)
        out = np.zeros(self.n, dtype=np.uint32)
        self.L.refh_expand_ternary(self.h, j, _p(pk, u8p), _p(out, u32p))
        return out

    def ntt_roots(self, j):
        r = np.zeros(self.n, dtype=np.uint32)
        self.L.refh_ntt_roots(self.h, j, _p(r, u32p))
        return r

    def ntt(self, vec, j):
        v = np.array(vec, dtype=np.uint32).copy()
        self.L.refh_ntt(self.h, j, _p(v, u32p))
        return v

    def reduce_pte(self, x, j):
        xin = np.ascontiguousarray(x, dtype=np.int64)
        out = np.zeros(self.n, dtype=np.uint32)
        self.L.refh_reduce_pte(self.h, j, _p(xin, i64p), _p(out, u32p))
        return out

    def reduce_e_small(self, e, j):
        ein = np.ascontiguousarray(e, dtype=np.int8)
        out = np.zeros(self.n, dtype=np.uint32)
        self.L.refh_reduce_e_small(self.h, j, _p(ein, i8p), _p(out, u32p))
        return out

    def intt(self, vec, j):
        v = np.array(vec, dtype=np.uint32).copy()
        self.L.refh_intt(self.h, j, _p(v, u32p))
        return v

    def decrypt(self, c0, c1, ntt_s, j):
        a = np.ascontiguousarray(c0, dtype=np.uint32)
        b = np.ascontiguousarray(c1, dtype=np.uint32)
        s = np.ascontiguousarray(ntt_s, dtype=np.uint32)
        out = np.zeros(self.n, dtype=np.uint32)
        self.L.refh_decrypt(self.h, j, _p(a, u32p), _p(b, u32p), _p(s, u32p), _p(out, u32p))
        return out

    def decode(self, pt, j, values_len=None):
        values_len = self.n // 2 if values_len is None else values_len
        x = np.ascontiguousarray(pt, dtype=np.uint32)
        out = np.zeros(values_len, dtype=np.float32)
        self.L.refh_decode(self.h, j, _p(x, u32p), values_len, _p(out, f32p))
        return out

    def barrett32(self, x, j=0):
        return int(self.L.refh_barrett32(self.h, j, x))

    def barrett64(self, x, j=0):
        return int(self.L.refh_barrett64(self.h, j, x))

    def mul_mod(self, a, b, j=0):
        return int(self.L.refh_mul_mod(self.h, j, a, b))

    def encrypt_sym(self, values, share_seed, seed):
        v = np.ascontiguousarray(values, dtype=np.float32).ravel()
        c0 = np.zeros((self.np, self.n), dtype=np.uint32)
        c1 = np.zeros((self.np, self.n), dtype=np.uint32)
        c1_alias = np.zeros((self.np, self.n), dtype=np.uint32)
        ntt_s = np.zeros((self.np, self.n), dtype=np.uint32)
        pte = np.zeros(self.n, dtype=np.int64)
        ctr = C.c_uint64(0)
        ok = self.L.refh_encrypt_sym(self.h, _p(v, f32p), v.size, _p(_seed(share_seed), u8p),
                                     _p(_seed(seed), u8p), _p(c0, u32p), _p(c1, u32p),
                                     _p(c1_alias, u32p), _p(pte, i64p), _p(ntt_s, u32p),
                                     C.byref(ctr))
        return dict(ok=bool(ok), c0=c0, c1=c1, c1_alias=c1_alias, pte=pte, ntt_s=ntt_s,
                    end_ctr=int(ctr.value))

    def encrypt_asym(self, values, seed, pk0, pk1):
        v = np.ascontiguousarray(values, dtype=np.float32).ravel()
        pk0 = np.ascontiguousarray(pk0, dtype=np.uint32)
        pk1 = np.ascontiguousarray(pk1, dtype=np.uint32)
        c0 = np.zeros((self.np, self.n), dtype=np.uint32)
        c1 = np.zeros((self.np, self.n), dtype=np.uint32)
        pte = np.zeros(self.n, dtype=np.int64)
        u = np.zeros(self.n // 4, dtype=np.uint8)
        e1 = np.zeros(self.n, dtype=np.int8)
        ctr = C.c_uint64(0)
        ok = self.L.refh_encrypt_asym(self.h, _p(v, f32p), v.size, _p(_seed(seed), u8p),
                                      _p(pk0, u32p), _p(pk1, u32p), _p(c0, u32p), _p(c1, u32p),
                                      _p(pte, i64p), _p(u, u8p), _p(e1, i8p), C.byref(ctr))
        return dict(ok=bool(ok), c0=c0, c1=c1, pte=pte, u=u, e1=e1, end_ctr=int(ctr.value))

    @classmethod
    def print_text(cls, path, name, poly=None, values=None):
        """The reference's print_poly_flpt_full / print_poly_full output, captured into `path`."""
        L = cls.lib()
        p = None if poly is None else np.ascontiguousarray(poly, dtype=np.uint32)
        v = None if values is None else np.ascontiguousarray(values, dtype=np.float32)
        L.refh_print_to_file(path.encode(), name.encode(), _p(p, u32p), 0 if p is None else p.size,
                             _p(v, f32p), 0 if v is None else v.size)

    @classmethod
    def gen_pk(cls, n, nprimes, sk_packed, pk_seed, ep_seed):
        L = cls.lib()
        sk = np.ascontiguousarray(sk_packed, dtype=np.uint8)
        pk0 = np.zeros((nprimes, n), dtype=np.uint32)
        pk1 = np.zeros((nprimes, n), dtype=np.uint32)
        L.refh_gen_pk(n, nprimes, _p(sk, u8p), _p(_seed(pk_seed), u8p), _p(_seed(ep_seed), u8p),
                      _p(pk0, u32p), _p(pk1, u32p))
        return pk0, pk1

    @classmethod
    def api_encrypt(cls, n, nprimes, asym, values, share_seed, seed):
        """se_setup + se_encrypt_seeded in the CWD (needs adapter_output_data/ files)."""
        L = cls.lib()
        v = np.ascontiguousarray(values, dtype=np.float32).ravel()
        cap = 8 * n * nprimes
        out = np.zeros(cap, dtype=np.uint8)
        ncalls = C.c_size_t(0)
        got = L.refh_api_encrypt(n, nprimes, 1 if asym else 0, _p(v, f32p), v.size * 4,
                                 _p(_seed(share_seed), u8p), _p(_seed(seed), u8p),
                                 _p(out, u8p), cap, C.byref(ncalls))
        return int(got), int(ncalls.value), out

    @classmethod
    def encrypt_sym_batch(cls, n, nprimes, values, share_seeds, seeds, sk_packed, nthreads=1,
                          keep=True):
        L = cls.lib()
        v = np.ascontiguousarray(values, dtype=np.float32)
        B = v.shape[0]
        ss = np.ascontiguousarray(share_seeds, dtype=np.uint8)
        sd = np.ascontiguousarray(seeds, dtype=np.uint8)
        sk = np.ascontiguousarray(sk_packed, dtype=np.uint8)
        c0 = np.zeros((B, nprimes, n), dtype=np.uint32) if keep else None
        c1 = np.zeros((B, nprimes, n), dtype=np.uint32) if keep else None
        L.refh_encrypt_sym_batch(n, nprimes, _p(v, f32p), B, _p(ss, u8p), _p(sd, u8p),
                                 _p(sk, u8p), _p(c0, u32p), _p(c1, u32p), nthreads)
        return c0, c1

    @classmethod
    def encrypt_asym_batch(cls, n, nprimes, values, seeds, pk0, pk1, nthreads=1, keep=True):
        L = cls.lib()
        v = np.ascontiguousarray(values, dtype=np.float32)
        B = v.shape[0]
        sd = np.ascontiguousarray(seeds, dtype=np.uint8)
        pk0 = np.ascontiguousarray(pk0, dtype=np.uint32)
        pk1 = np.ascontiguousarray(pk1, dtype=np.uint32)
        c0 = np.zeros((B, nprimes, n), dtype=np.uint32) if keep else None
        c1 = np.zeros((B, nprimes, n), dtype=np.uint32) if keep else None
        L.refh_encrypt_asym_batch(n, nprimes, _p(v, f32p), B, _p(sd, u8p), _p(pk0, u32p), _p(pk1, u32p),
                                  _p(c0, u32p), _p(c1, u32p), nthreads)
        return c0, c1

    @classmethod
    def encode_ntt_batch(cls, n, nprimes, values, nthreads=1, keep=True):
        L = cls.lib()
        v = np.ascontiguousarray(values, dtype=np.float32)
        B = v.shape[0]
        out = np.zeros((B, nprimes, n), dtype=np.uint32) if keep else None
        L.refh_encode_ntt_batch(n, nprimes, _p(v, f32p), B, _p(out, u32p), nthreads)
        return out

    @classmethod
    def api_print_lines(cls, n, nprimes, asym, values, share_seed, seed, path):
        """se_encrypt_seeded(print=true) of the reference: the "c0: " / "c1: " lines it prints."""
        L = cls.lib()
        v = np.ascontiguousarray(values, dtype=np.float32)
        ok = L.refh_api_encrypt_print(n, nprimes, 1 if asym else 0, _p(v, f32p), v.nbytes, _p(_seed(share_seed), u8p),
                                      _p(_seed(seed), u8p), path.encode())
        assert ok
        return [l for l in open(path).read().splitlines(True) if l.startswith(("c0: ", "c1: "))]
