"""GPU parity tests (-m gpu) of the reference-named LOWER surface (include/seal_embedded_amd_lower.h):
callers written against the reference's own prototypes -- plain-gcc C programs in the shape of
device/test/ckks_tests_sym.c:103-172 / ckks_tests_asym.c:120-208 and direct ctypes calls -- get
results bit-identical to the golden vectors of the compiled reference and to the oracle.
Nothing here reads /root/reference.
"""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np
import pytest

import vectors as V

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

SEED_PK = hashlib.shake_256(b"golden-pk").digest(64)
SEED_EP = hashlib.shake_256(b"golden-ep").digest(64)


@pytest.fixture(scope="module")
def env():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (no CPU fallback exists)")
    import __graft_entry__ as ge
    pkg = ge.load_package()
    from oracle import pyoracle
    pyoracle.build(ref=False)
    return dict(torch=torch, pkg=pkg)


def _build_caller(name, tmp_path):
    exe = tmp_path / name
    lib = os.path.join(ROOT, "seal-embedded_amd", "lib")
    inc = os.path.join(ROOT, "include")
    subprocess.run(["gcc", "-std=gnu11", "-Wall", "-Wextra", "-Werror",
                    os.path.join(ROOT, "tests", "c", name + ".c"), "-I" + inc,
                    "-I" + os.path.join(inc, "compat"), "-L" + lib, "-lseal_embedded_amd",
                    "-Wl,-rpath," + lib, "-o", str(exe)], check=True)
    return exe


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("shape", V.ALL_SHAPES, ids=lambda s: f"{s[0]}x{s[1]}")
def test_reference_style_sym_caller_matches_golden(env, golden, tmp_path, shape):
    """tests/c/lower_sym_caller.c (reference API names only, gcc) -> per-prime c0, c1_save, s_save,
    the aliased c1 buffer, m + e and the final PRNG counter equal the compiled reference's."""
    n, npr = shape
    exe = _build_caller("lower_sym_caller", tmp_path)
    data = tmp_path / "adapter_output_data"
    data.mkdir()
    V.secret_key(n).tofile(data / f"sk_{n}.dat")
    out = tmp_path / "out.bin"
    subprocess.run([str(exe), str(n), str(npr), str(out)], cwd=tmp_path, check=True, timeout=300)
    raw = out.read_bytes()
    assert len(raw) == 8 * n + npr * 4 * 4 * n + 8
    pte = np.frombuffer(raw, dtype=np.int64, count=n)
    polys = np.frombuffer(raw, dtype=np.uint32, offset=8 * n, count=npr * 4 * n).reshape(npr, 4, n)
    end_ctr = int(np.frombuffer(raw[-8:], dtype=np.uint64)[0])
    g = golden["digests"]["shapes"][f"{n}x{npr}"]["sym_survey"]
    assert _sha(pte) == g["pte_sha256"]
    assert _sha(polys[:, 0]) == g["c0_sha256"]
    assert _sha(polys[:, 1]) == g["c1_sha256"]          # c1_save = a
    assert _sha(polys[:, 2]) == g["ntt_s_sha256"]
    assert _sha(polys[:, 3]) == g["c1_alias_sha256"]    # c1_ptr == ntt_pte_ptr: NTT(m + e)
    assert end_ctr == g["end_ctr"]
    for j in range(npr):
        assert [int(x) for x in polys[j, 0, :8]] + [int(x) for x in polys[j, 0, -8:]] == g["c0_ends"][j]


@pytest.mark.parametrize("shape", [(1024, 1), (4096, 3), (16384, 6)], ids=lambda s: f"{s[0]}x{s[1]}")
def test_reference_style_asym_caller_matches_golden(env, golden, tmp_path, shape):
    """tests/c/lower_asym_caller.c: gen_pk per prime, ckks_asym_init, ckks_encode_encrypt_asym."""
    from oracle.pyoracle import Oracle
    n, npr = shape
    exe = _build_caller("lower_asym_caller", tmp_path)
    data = tmp_path / "adapter_output_data"
    data.mkdir()
    sk = V.secret_key(n)
    sk.tofile(data / f"sk_{n}.dat")
    (tmp_path / "pk_seed.bin").write_bytes(SEED_PK)
    (tmp_path / "ep_seed.bin").write_bytes(SEED_EP)
    out = tmp_path / "out.bin"
    subprocess.run([str(exe), str(n), str(npr), str(out), "pk_seed.bin", "ep_seed.bin"], cwd=tmp_path,
                   check=True, timeout=300)
    raw = out.read_bytes()
    off = 0

    def take(dtype, count):
        nonlocal off
        a = np.frombuffer(raw, dtype=dtype, offset=off, count=count)
        off += a.nbytes
        return a
    pk0, pk1 = take(np.uint32, npr * n), take(np.uint32, npr * n)
    pte, u, e1 = take(np.int64, n), take(np.uint8, n // 4), take(np.int8, n)
    per = take(np.uint32, npr * 5 * n).reshape(npr, 5, n)
    ctr = int(take(np.uint64, 1)[0])
    assert off == len(raw)
    g = golden["digests"]["shapes"][f"{n}x{npr}"]["asym_survey"]
    assert _sha(pk0) == g["pk0_sha256"] and _sha(pk1) == g["pk1_sha256"]
    assert _sha(pte) == g["pte_sha256"] and _sha(u) == g["u_sha256"] and _sha(e1) == g["e1_sha256"]
    assert _sha(per[:, 0]) == g["c0_sha256"] and _sha(per[:, 1]) == g["c1_sha256"]
    assert ctr == g["end_ctr"]
    # the test-only saves against the oracle: NTT(expand(u)), NTT(e1), NTT(m + e0)
    o = Oracle(n, npr)
    for j in range(npr):
        assert (per[j, 2] == o.ntt(o.expand_ternary(u, j), j)).all()
        assert (per[j, 3] == o.ntt(o.reduce_e_small(e1, j), j)).all()
        assert (per[j, 4] == o.ntt(o.reduce_pte(pte, j), j)).all()


# ------------------------------------------------------------------------------------ ctypes
class Modulus(C.Structure):
    _fields_ = [("value", C.c_uint32), ("const_ratio", C.c_uint32 * 2)]


class Parms(C.Structure):
    _fields_ = [("coeff_count", C.c_size_t), ("logn", C.c_size_t), ("moduli", C.POINTER(Modulus)),
                ("curr_modulus", C.POINTER(Modulus)), ("curr_modulus_idx", C.c_size_t),
                ("nprimes", C.c_size_t), ("scale", C.c_double), ("is_asymmetric", C.c_bool),
                ("pk_from_file", C.c_bool), ("sample_s", C.c_bool), ("small_s", C.c_bool),
                ("small_u", C.c_bool)]


class Prng(C.Structure):
    _fields_ = [("seed", C.c_uint8 * 64), ("counter", C.c_uint64)]


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def _lower(env):
    L = env["pkg"].lib()
    L.next_modulus.restype = C.c_bool
    L.ckks_encode_base.restype = C.c_bool
    L.ckks_get_mempool_size_sym.restype = C.c_size_t
    L.ckks_get_mempool_size_asym.restype = C.c_size_t
    return L


def _prng(seed, ctr=0):
    p = Prng()
    C.memmove(p.seed, bytes(seed), 64)
    p.counter = ctr
    return p


@pytest.mark.parametrize("shape", [(1024, 1), (4096, 3), (8192, 6)], ids=lambda s: f"{s[0]}x{s[1]}")
def test_lower_stage_functions_vs_oracle(env, shape):
    """Every stage of the lower surface, called with the reference's prototypes through ctypes, on
    random operands against the oracle: parameters, index map, root tables, ntt/intt, ifft/fft,
    reduce_*, expand, the four samplers with carried counters, prng_fill_buffer."""
    from oracle.pyoracle import Oracle
    L = _lower(env)
    n, npr = shape
    o = Oracle(n, npr)
    rng = np.random.default_rng(n + npr)
    P = Parms()
    imap = np.zeros(n, np.uint16)
    L.ckks_setup(C.c_size_t(n), C.c_size_t(npr), _vp(imap), C.byref(P))
    assert P.coeff_count == n and P.nprimes == npr and P.scale == o.p.scale and P.curr_modulus_idx == 0
    assert (imap == o.map).all()
    assert [P.moduli[j].value for j in range(npr)] == [int(o.p.q[j]) for j in range(npr)]
    assert [(P.moduli[j].const_ratio[0], P.moduli[j].const_ratio[1]) for j in range(npr)] == \
        [(int(o.p.cr_lo[j]), int(o.p.cr_hi[j])) for j in range(npr)]

    # FFT family (complex128, in place)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex128)
    y = x.copy()
    L.ifft_inpl(_vp(y), C.c_size_t(n), C.c_size_t(P.logn), None)
    assert y.tobytes() == o.ifft(x).tobytes()
    z = x.copy()
    L.fft_inpl(_vp(z), C.c_size_t(n), C.c_size_t(P.logn), None)
    assert z.tobytes() == o.fft(x).tobytes()
    roots = np.zeros(n, np.complex128)
    L.calc_ifft_roots(C.c_size_t(n), C.c_size_t(P.logn), _vp(roots))
    assert roots.tobytes() == o.twiddles().tobytes()

    seed = V.derive_seeds(f"lower-{n}", 3)
    for j in range(npr):
        q = int(o.p.q[j])
        r = np.zeros(n, np.uint32)
        L.ntt_roots_initialize(C.byref(P), _vp(r))
        assert (r == o.ntt_roots(j)).all()
        v = rng.integers(0, q, n, dtype=np.uint64).astype(np.uint32)
        w = v.copy()
        L.ntt_inpl(C.byref(P), _vp(r), _vp(w))
        assert (w == o.ntt(v, j)).all()
        ir = np.zeros(n, np.uint32)
        L.intt_roots_initialize(C.byref(P), _vp(ir))
        L.intt_inpl(C.byref(P), _vp(ir), _vp(w))
        assert (w == v).all() and (o.intt(o.ntt(v, j), j) == v).all()
        # reduce_* incl. negative multiples of q (-> non-canonical q) and extreme magnitudes
        m = rng.integers(-2 ** 62, 2 ** 62, n, dtype=np.int64)
        m[:6] = [0, -q, q, -1, 1, -(2 ** 62)]
        red = np.zeros(n, np.uint32)
        L.reduce_set_pte(C.byref(P), _vp(m), _vp(red))
        assert (red == o.reduce_pte(m, j)).all() and red[1] == q
        acc = v.copy()
        L.reduce_add_pte(C.byref(P), _vp(m), _vp(acc))
        exp = (v.astype(np.uint64) + o.reduce_pte(m, j).astype(np.uint64))
        exp = np.where(exp >= q, exp - q, exp)
        assert (acc == exp.astype(np.uint32)).all()
        e = rng.integers(-21, 22, n).astype(np.int8)
        L.reduce_set_e_small(C.byref(P), _vp(e), _vp(red))
        assert (red == o.reduce_e_small(e, j)).all()
        # samplers with the counter carried through the SE_PRNG object
        pr = _prng(seed[0], 5 * j)
        a = np.zeros(n, np.uint32)
        L.sample_poly_uniform(C.byref(P), C.byref(pr), _vp(a))
        ea, ectr = o.sample_uniform(j, seed[0].tobytes(), 5 * j)
        assert (a == ea).all() and pr.counter == ectr
        packed = V.secret_key(n, seed=3 + j)
        ex = np.zeros(n, np.uint32)
        L.expand_poly_ternary(_vp(packed), C.byref(P), _vp(ex))
        assert (ex == o.expand_ternary(packed, j)).all()
        if j + 1 < npr:
            assert L.next_modulus(C.byref(P))
    assert not L.next_modulus(C.byref(P)) and P.curr_modulus_idx == 0   # wraps at the chain's end

    pr = _prng(seed[1], 7)
    u = np.zeros(n // 4, np.uint8)
    L.sample_small_poly_ternary_prng_96(C.c_size_t(n), C.byref(pr), _vp(u))
    eu, ectr = o.sample_ternary_small(seed[1].tobytes(), 7)
    assert (u == eu).all() and pr.counter == ectr
    e8 = np.zeros(n, np.int8)
    L.sample_poly_cbd_generic_prng_16(C.c_size_t(n), C.byref(pr), _vp(e8))
    ee, ectr2 = o.cbd_int8(seed[1].tobytes(), ectr)
    assert (e8 == ee).all() and pr.counter == ectr2
    m = rng.integers(-2 ** 40, 2 ** 40, n, dtype=np.int64)
    m2 = m.copy()
    L.sample_add_poly_cbd_generic_inpl_prng_16(_vp(m2), C.c_size_t(n), C.byref(pr))
    em, ectr3 = o.cbd_add(m, seed[1].tobytes(), ectr2)
    assert (m2 == em).all() and pr.counter == ectr3
    for ln in (1, 4, 96, 137, 4 * n):
        pr = _prng(seed[2], 2 ** 40 + ln)
        buf = np.zeros(ln, np.uint8)
        L.prng_fill_buffer(C.c_size_t(ln), C.byref(pr), _vp(buf))
        assert buf.tobytes() == o.prng_block(seed[2].tobytes(), 2 ** 40 + ln, ln)
        assert pr.counter == 2 ** 40 + ln + 1
    L.delete_parameters(C.byref(P))


def test_lower_encode_base_contract(env, golden):
    """ckks_encode_base with the reference's memory contract: the nine reference patterns against
    the golden digests; a short values_len leaves the unreached slots of conj_vals as the caller had
    them; an overflowing input returns false and leaves the buffer converted up to the first bad
    index exactly as the reference's in-place loop does (ckks_common.c:187-206)."""
    from oracle.pyoracle import Oracle
    L = _lower(env)
    n, npr = 4096, 3
    o = Oracle(n, npr)
    P = Parms()
    imap = np.zeros(n, np.uint16)
    L.ckks_setup(C.c_size_t(n), C.c_size_t(npr), _vp(imap), C.byref(P))
    g = golden["digests"]["shapes"][f"{n}x{npr}"]["encode"]
    conj = np.zeros(n, np.complex128)
    for t in range(9):
        v = V.pattern_values(t, n)
        conj[:] = 0
        assert L.ckks_encode_base(C.byref(P), _vp(v), C.c_size_t(n // 2), _vp(imap), None, _vp(conj))
        assert _sha(conj.view(np.int64)[:n]) == g[f"pattern{t}"]["sha256"]
    # values_len < n/2: stale slots take part in the transform
    rng = np.random.default_rng(5)
    stale = (rng.integers(-50, 50, n).astype(np.float64) + 0j).astype(np.complex128)
    v = V.bench_values(1, n)[0]
    k = 100
    conj[:] = stale
    assert L.ckks_encode_base(C.byref(P), _vp(v), C.c_size_t(k), _vp(imap), None, _vp(conj))
    x = stale.copy()
    x[imap[:k]] = v[:k].astype(np.float64)
    x[imap[n // 2:n // 2 + k]] = v[:k].astype(np.float64)
    y = o.ifft(x).real * (o.p.scale / n)                  # C round(): half away from zero
    exp = np.where(y >= 0, np.floor(y + 0.5), np.ceil(y - 0.5)).astype(np.int64)
    assert (conj.view(np.int64)[:n] == exp).all()
    # overflow: false, and the in-place conversion stopped at the first failing index
    big = np.full(n // 2, 3.0e38, dtype=np.float32)
    conj[:] = 0
    assert not L.ckks_encode_base(C.byref(P), _vp(big), C.c_size_t(n // 2), _vp(imap), None, _vp(conj))
    assert not g["overflow_3e38_ok"]
    x = np.zeros(n, np.complex128)
    x[imap[:n // 2]] = float(big[0])          # the float32 value, widened (ckks_common.c:148)
    x[imap[n // 2:]] = float(big[0])
    full = o.ifft(x)
    coeff = full.real * (o.p.scale / n)
    bad = int(np.argmax(np.abs(np.where(coeff >= 0, np.floor(coeff + 0.5), np.ceil(coeff - 0.5))) > 2.0 ** 63))
    ref = full.copy().view(np.uint8)
    ints = np.where(coeff >= 0, np.floor(coeff + 0.5), np.ceil(coeff - 0.5))[:bad].astype(np.int64)
    ref[:8 * bad] = ints.view(np.uint8)
    assert conj.view(np.uint8).tobytes() == ref.tobytes()
    L.delete_parameters(C.byref(P))


@pytest.mark.parametrize("shape", [(1024, 1), (4096, 3), (16384, 6)], ids=lambda s: f"{s[0]}x{s[1]}")
def test_lower_encode_base_nonfinite_values(env, golden, shape):
    """ckks_encode_base / ifft_inpl under the reference's names on NaN / Inf / FLT_MAX / subnormal / -0.0
    values: the return value, the index at which the in-place conversion stopped and every int64 before it
    equal the compiled reference's (goldens) and the oracle's; the unconverted tail is the IFFT output, equal
    to the oracle's up to the sign / payload bits of NaNs (x86 produces the negative default NaN, the GPU the
    positive one -- a NaN never reaches an integer output except as INT64_MIN)."""
    from oracle.pyoracle import Oracle
    L = _lower(env)
    L.ifft_inpl.restype = None
    n, npr = shape
    o = Oracle(n, npr)
    P = Parms()
    imap = np.zeros(n, np.uint16)
    L.ckks_setup(C.c_size_t(n), C.c_size_t(npr), _vp(imap), C.byref(P))
    gold = golden["digests"]["shapes"][f"{n}x{npr}"]["encode"]["nonfinite"]
    conj = np.zeros(n, np.complex128)
    mids = 0
    for c in range(V.NONFINITE_CASES):
        v = V.nonfinite_values(c, n)
        conj[:] = 0
        ok = bool(L.ckks_encode_base(C.byref(P), _vp(v), C.c_size_t(n // 2), _vp(imap), None, _vp(conj)))
        idx, m = o.encode_ex(v)
        assert idx == gold[c]["fail_index"] and ok == (idx == n), (c, ok, idx)
        got = conj.view(np.int64)
        assert (got[:idx] == m[:idx]).all(), c
        assert V.sha256_hex(got[:idx]) == gold[c]["prefix_sha256"], c
        mids += 0 < idx < n
        # tail: raw IFFT output from complex element ceil(idx / 2) on
        x = np.zeros(n, np.complex128)
        x[imap[:n // 2]] = v.astype(np.float64)
        x[imap[n // 2:]] = v.astype(np.float64)
        full = o.ifft(x)
        k0 = (idx + 1) // 2
        assert np.array_equal(conj[k0:].view(np.float64), full[k0:].view(np.float64), equal_nan=True), c
        # the stand-alone operator on the same (complex) input
        y = x.copy()
        L.ifft_inpl(_vp(y), C.c_size_t(n), C.c_size_t(int(np.log2(n))), None)
        assert np.array_equal(y.view(np.float64), full.view(np.float64), equal_nan=True), c
    assert mids >= 2
    L.delete_parameters(C.byref(P))


def test_pool_carving_is_the_reference_default_layout(env):
    """ckks_set_ptrs_sym / _asym: offsets of the default configuration (ckks_sym.c:78-160,
    ckks_asym.c:75-157), incl. the c1 == ntt_pte alias of the symmetric pool."""
    L = _lower(env)

    class Ptrs(C.Structure):
        _fields_ = [(k, C.c_void_p) for k in ("conj_vals", "ifft_roots", "values", "ternary",
                                              "conj_vals_int_ptr", "c0_ptr", "c1_ptr", "index_map_ptr",
                                              "ntt_roots_ptr", "ntt_pte_ptr", "e1_ptr")]
    n = 4096
    for asym in (False, True):
        size = (L.ckks_get_mempool_size_asym if asym else L.ckks_get_mempool_size_sym)(C.c_size_t(n))
        pool = np.zeros(size, np.uint32)
        p = Ptrs()
        (L.ckks_set_ptrs_asym if asym else L.ckks_set_ptrs_sym)(C.c_size_t(n), _vp(pool), C.byref(p))
        base = pool.ctypes.data
        off = {k: (getattr(p, k) - base) // 4 if getattr(p, k) else None for k, _ in Ptrs._fields_}
        assert off["conj_vals"] == 0 and off["conj_vals_int_ptr"] == 0 and off["ifft_roots"] is None
        assert off["c1_ptr"] == 2 * n and off["c0_ptr"] == 3 * n and off["ntt_roots_ptr"] == 4 * n
        if asym:
            assert size == 4 * n + (n + n // 4 + n // 16) + n + n // 2 + n // 2
            assert off["ntt_pte_ptr"] == 5 * n and off["index_map_ptr"] == 6 * n
            assert off["e1_ptr"] == 6 * n + n // 2 and off["ternary"] == 6 * n + n // 2 + n // 4
            assert off["values"] == 6 * n + n // 2 + n // 4 + n // 16
        else:
            assert size == 4 * n + n + n // 2 + n // 16 + n // 2
            assert off["ntt_pte_ptr"] == off["c1_ptr"]
            assert off["index_map_ptr"] == 5 * n and off["ternary"] == 5 * n + n // 2
            assert off["values"] == 5 * n + n // 2 + n // 16
        assert max(v for v in off.values() if v is not None) < size


@pytest.mark.parametrize("shape", [(4096, 3), (8192, 6), (1024, 1)], ids=lambda s: f"{s[0]}x{s[1]}")
def test_lower_sym_fast_path_keeps_the_reference_bytes_when_inputs_move(env, shape, monkeypatch):
    """Round 5: ckks_sym_init (ckks_sym.c:181-197) starts the uniform samplers of ALL primes asynchronously (prime
    speculation), the first ckks_encode_encrypt_sym (ckks_sym.c:199-301) computes every prime of the ciphertext, later
    per-prime calls return their prime from that -- IF their inputs still match.  Every call here is compared with what
    the reference computes from the inputs of THAT call: the plain sequence; the shareable PRNG's counter moved between
    two primes (inside and far outside the speculated window); a plaintext coefficient changed between two primes; the
    key changed; the PRNG re-seeded; a prime repeated; the gen_pk form (ep_small) in between; and the whole sequence
    again with SE_AMD_LOWER_SPECULATION=0 (same bytes)."""
    from oracle.pyoracle import Oracle
    L = _lower(env)
    n, npr = shape
    o = Oracle(n, npr)
    rng = np.random.default_rng(7 * n + npr)
    seeds = V.derive_seeds(f"lower-fast-{n}", 4)
    sk = V.secret_key(n, seed=29)
    sk2 = V.secret_key(n, seed=31)

    def one_prime(P, pte, pr, key, j, ep=None):
        """call + check of one ckks_encode_encrypt_sym against the oracle's per-stage functions"""
        c0, c1, ntt_pte = (np.zeros(n, np.uint32) for _ in range(3))
        s_save, c1_save, roots = (np.zeros(n, np.uint32) for _ in range(3))
        start = int(pr.counter)
        sd = bytes(pr.seed)
        L.ckks_encode_encrypt_sym(C.byref(P), None if ep is not None else _vp(pte), None if ep is None else _vp(ep),
                                  C.byref(pr), _vp(key), _vp(ntt_pte), _vp(roots), _vp(c0), _vp(c1), _vp(s_save),
                                  _vp(c1_save))
        a, ectr = o.sample_uniform(j, sd, start)
        q = int(o.p.q[j])
        s_hat = o.ntt(o.expand_ternary(key, j), j)
        x = o.ntt(o.reduce_pte(pte, j) if ep is None else o.reduce_e_small(ep, j), j)
        ec0 = (x.astype(np.uint64) + q - (a.astype(np.uint64) * s_hat.astype(np.uint64)) % q) % q
        assert pr.counter == ectr, (j, start)
        assert (c1 == a).all() and (c1_save == a).all(), (j, start)
        assert (s_save == s_hat).all() and (ntt_pte == x).all() and (roots == o.ntt_roots(j)).all(), (j, start)
        assert (c0 == ec0.astype(np.uint32)).all(), (j, start)

    def sequence():
        P = Parms()
        imap = np.zeros(n, np.uint16)
        L.ckks_setup(C.c_size_t(n), C.c_size_t(npr), _vp(imap), C.byref(P))
        m = rng.integers(-2 ** 30, 2 ** 30, n, dtype=np.int64)

        def init(seed_a, seed_e):
            pte = m.copy()
            pa, pe = Prng(), Prng()
            sa = np.frombuffer(bytes(seed_a), np.uint8).copy()
            se = np.frombuffer(bytes(seed_e), np.uint8).copy()
            L.ckks_sym_init(C.byref(P), _vp(sa), _vp(se), C.byref(pa), C.byref(pe), _vp(pte))
            em, ectr = o.cbd_add(m, bytes(seed_e), 0)
            assert (pte == em).all() and pe.counter == ectr and pa.counter == 0 and bytes(pa.seed) == bytes(seed_a)
            return pte, pa

        def rewind():
            while P.curr_modulus_idx != 0:
                L.next_modulus(C.byref(P))

        # 1. the plain sequence, twice (the second init re-arms the table while the first one's results are around)
        for rep in range(2):
            pte, pa = init(seeds[rep], seeds[2])
            for j in range(npr):
                one_prime(P, pte, pa, sk, j)
                if j + 1 < npr:
                    assert L.next_modulus(C.byref(P))
            rewind()
        if npr == 1:
            pte, pa = init(seeds[0], seeds[2])
            pa.counter = 3                       # not the counter the table was built for
            one_prime(P, pte, pa, sk, 0)
            one_prime(P, pte, pa, sk2, 0)         # a repeated prime, another key, the counter where it stands
            L.delete_parameters(C.byref(P))
            return
        # 2. the counter moves between primes: by a little (still inside prime 1's window) and by a lot
        for bump in (1, 2, 10 ** 6):
            pte, pa = init(seeds[0], seeds[3])
            one_prime(P, pte, pa, sk, 0)
            L.next_modulus(C.byref(P))
            pa.counter += bump
            for j in range(1, npr):
                one_prime(P, pte, pa, sk, j)
                if j + 1 < npr:
                    L.next_modulus(C.byref(P))
            rewind()
        # 2b. the reference's bench order (device/bench/bench_sym.c:96-145 never rewinds the Parms between its
        #     iterations): the init call finds the Parms at the LAST prime, the chain runs np-1, 0, 1, ...
        for start in (npr - 1, 1):
            rewind()
            for _ in range(start):
                L.next_modulus(C.byref(P))
            pte, pa = init(seeds[2], seeds[0])
            for k in range(npr):
                one_prime(P, pte, pa, sk, (start + k) % npr)
                L.next_modulus(C.byref(P))
            rewind()
        # 3. the plaintext, then the key, change between primes; a prime is repeated; gen_pk's form in between
        pte, pa = init(seeds[1], seeds[3])
        one_prime(P, pte, pa, sk, 0)
        L.next_modulus(C.byref(P))
        pte[5] += 12345
        pte[n - 1] = -pte[n - 1] - 1
        one_prime(P, pte, pa, sk, 1)
        keep = int(pa.counter)
        ep = rng.integers(-21, 22, n).astype(np.int8)
        pk = _prng(seeds[2], 0)
        one_prime(P, pte, pk, sk, 1, ep=ep)                   # ckks_sym.c:279-283: ep_small wins, own PRNG
        assert pa.counter == keep
        if npr > 2:
            L.next_modulus(C.byref(P))
            one_prime(P, pte, pa, sk2, 2)                     # another key from here on
            back = _prng(seeds[1], keep)
            one_prime(P, pte, back, sk2, 2)                   # the same prime again, same inputs
            one_prime(P, pte, _prng(seeds[0], keep), sk2, 2)  # ... and under another seed
        rewind()
        L.delete_parameters(C.byref(P))

    sequence()
    monkeypatch.setenv("SE_AMD_LOWER_SPECULATION", "0")
    sequence()
