"""Formats on either side of the path, host-only, no GPU needed:
adapter text lines pinned against the reference's own printers (golden text produced by
util_print.h through oracle/_ref), SEAL uint64 layout, key files."""
import ctypes as C
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def L():
    import __graft_entry__ as ge
    pkg = ge.load_package()
    pkg.build_library()
    return pkg.lib()


def _fmt(L, fn, name, arr):
    need = fn(name.encode(), arr.ctypes.data_as(C.c_void_p), arr.size, None, 0)
    buf = C.create_string_buffer(need + 1)
    got = fn(name.encode(), arr.ctypes.data_as(C.c_void_p), arr.size, buf, need + 1)
    assert got == need
    return buf.value.decode()


def parse_adapter_line(text):
    """What adapter/fileops.h:220-287 (poly_string_file_load) does with one line."""
    body = text[text.index("{") + 1:]
    vals = []
    for tok in body.split():
        if "}" in tok:
            break
        vals.append(tok.replace(",", ""))
    return vals


def test_text_lines_match_reference_printers(L, golden):
    g = golden["digests"]["text_format"]
    poly = np.array(g["poly"], dtype=np.uint32)
    vals = np.array(g["values"], dtype=np.float32)
    assert _fmt(L, L.se_amd_format_poly_text, "c0", poly) == g["poly_line"]
    assert _fmt(L, L.se_amd_format_values_text, "v (cleartext)", vals) == g["values_line"]
    assert [int(x) for x in parse_adapter_line(g["poly_line"])] == g["poly"]


def test_text_lines_vs_live_reference(L, tmp_path):
    from oracle import pyoracle
    if not pyoracle.ref_available():
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(3)
    poly = rng.integers(0, 2 ** 32, 4096, dtype=np.uint64).astype(np.uint32)
    vals = (rng.standard_normal(2048) * 50).astype(np.float32)
    f = str(tmp_path / "t.txt")
    pyoracle.Reference.print_text(f, "c1", poly=poly, values=vals)
    want = open(f).read()
    got = (_fmt(L, L.se_amd_format_values_text, "v (cleartext)", vals) +
           _fmt(L, L.se_amd_format_poly_text, "c1", poly))
    assert got == want


def test_ciphertext_text_file_order_and_roundtrip(L, tmp_path):
    n, npr = 1024, 1
    rng = np.random.default_rng(4)
    c0 = rng.integers(0, 134012929, (2, npr, n), dtype=np.uint64).astype(np.uint32)
    c1 = rng.integers(0, 134012929, (2, npr, n), dtype=np.uint64).astype(np.uint32)
    vals = rng.standard_normal((2, n // 2)).astype(np.float32)
    path = str(tmp_path / "ct.txt").encode()
    for b in range(2):
        rc = L.se_amd_write_ciphertext_text(path, 1 if b else 0, vals[b].ctypes.data_as(C.c_void_p), n // 2,
                                            c0[b].ctypes.data_as(C.c_void_p),
                                            c1[b].ctypes.data_as(C.c_void_p), n, npr)
        assert rc == 0
    lines = open(path.decode()).read().splitlines()
    assert len(lines) == 2 * (1 + 2 * npr)
    assert lines[0].startswith("v (cleartext) : { ") and lines[1].startswith("c0 : { ")
    assert lines[2].startswith("c1 : { ") and lines[3].startswith("v (cleartext)")
    assert [int(x) for x in parse_adapter_line(lines[4])] == [int(x) for x in c0[1, 0]]
    assert [int(x) for x in parse_adapter_line(lines[5])] == [int(x) for x in c1[1, 0]]


def test_seal_layout_packer(L):
    n, npr = 16, 3
    c0 = np.arange(npr * n, dtype=np.uint32).reshape(npr, n) + 1000
    c1 = np.arange(npr * n, dtype=np.uint32).reshape(npr, n) + 5000
    out = np.zeros(2 * npr * n, dtype=np.uint64)
    L.se_amd_pack_seal_ciphertext_host(c0.ctypes.data_as(C.c_void_p), c1.ctypes.data_as(C.c_void_p), n, npr,
                                       out.ctypes.data_as(C.c_void_p))
    for j in range(npr):                      # adapter/fileops.cpp:515-527
        assert (out[j * n:(j + 1) * n] == c0[j]).all()
        assert (out[npr * n + j * n:npr * n + (j + 1) * n] == c1[j]).all()


def test_key_files_roundtrip(L, tmp_path):
    import vectors as V
    n, npr = 4096, 3
    q = np.array([1053818881, 1054015489, 1054212097], dtype=np.uint32)
    sk = V.secret_key(n)
    rng = np.random.default_rng(8)
    pk0 = rng.integers(0, 1053818881, (npr, n), dtype=np.uint64).astype(np.uint32)
    pk1 = rng.integers(0, 1053818881, (npr, n), dtype=np.uint64).astype(np.uint32)
    d = str(tmp_path).encode()
    assert L.se_amd_save_secret_key_file(d, n, sk.ctypes.data_as(C.c_void_p)) == 0
    assert L.se_amd_save_public_key_files(d, n, npr, q.ctypes.data_as(C.c_void_p),
                                          pk0.ctypes.data_as(C.c_void_p), pk1.ctypes.data_as(C.c_void_p)) == 0
    assert (np.fromfile(tmp_path / f"sk_{n}.dat", dtype=np.uint8) == sk).all()      # fileops.c:140-170
    for j in range(npr):                                                             # fileops.c:172-204
        assert (np.fromfile(tmp_path / f"pk0_ntt_{n}_{q[j]}.dat", dtype=np.uint32) == pk0[j]).all()
        assert (np.fromfile(tmp_path / f"pk1_ntt_{n}_{q[j]}.dat", dtype=np.uint32) == pk1[j]).all()


@pytest.mark.parametrize("shape,mode", [((1024, 1), "sym"), ((4096, 3), "sym"), ((4096, 3), "asym"),
                                        ((16384, 6), "sym")], ids=["C1", "C2", "C3", "C4"])
def test_adapter_reader_roundtrip(L, tmp_path, shape, mode):
    """SEAL-side closure as far as this image allows: ciphertexts of the four encrypting
    BASELINE shapes, written by the product's text writer in the adapter's test order (values line, then per
    prime a c0 and a c1 line; device/test/api_tests.c:30-42,75-90), are read back by a restatement of the
    adapter's own reader (tests/adapter_reader.py: poly_string_file_load / ct_string_file_load with the file
    position threaded through three consecutive tests) and reconstruct exactly the uint64
    [component][prime][coeff] record the SEAL-layout packer builds -- and the values the writer printed
    parse back to the floats within the printer's precision."""
    import vectors as V
    from adapter_reader import read_tests
    from oracle.pyoracle import Oracle
    n, npr = shape
    o = Oracle(n, npr)
    sk = V.secret_key(n)
    ntests = 3
    vals = V.bench_values(ntests, n, first=5)
    ss, sd = V.bench_seeds(ntests, first=5)
    if mode == "asym":
        pk0, pk1 = o.gen_pk(sk, bytes(64), bytes(range(64)))
    path = str(tmp_path / "ct.txt").encode()
    want = []
    for t in range(ntests):
        r = (o.encrypt_sym(vals[t], ss[t].tobytes(), sd[t].tobytes(), sk) if mode == "sym"
             else o.encrypt_asym(vals[t], sd[t].tobytes(), pk0, pk1))
        c0, c1 = np.ascontiguousarray(r["c0"]), np.ascontiguousarray(r["c1"])
        assert L.se_amd_write_ciphertext_text(path, 1 if t else 0, vals[t].ctypes.data_as(C.c_void_p), n // 2,
                                              c0.ctypes.data_as(C.c_void_p), c1.ctypes.data_as(C.c_void_p),
                                              n, npr) == 0
        packed = np.zeros(2 * npr * n, dtype=np.uint64)
        L.se_amd_pack_seal_ciphertext_host(c0.ctypes.data_as(C.c_void_p), c1.ctypes.data_as(C.c_void_p), n, npr,
                                           packed.ctypes.data_as(C.c_void_p))
        want.append(packed)
    got = read_tests(open(path.decode(), "rb").read(), n, npr, ntests)
    q = np.array(o.q, dtype=np.uint64)
    for t in range(ntests):
        v, ct = got[t]
        assert np.array_equal(ct, want[t]), t
        assert v.shape == (n // 2,) and np.abs(v - vals[t].astype(np.float64)).max() < 1e-4
        # what SEAL's context checks on load: every residue below its prime (is_data_valid_for)
        assert (ct.reshape(2, npr, n) < q[None, :, None]).all()


def test_adapter_reader_token_rules():
    """The reader's corner rules the writer must respect: a '}' glued to the last value would DROP that value
    (the token containing '}' ends the object unread), commas are stripped, the object name is ignored, and a
    second call continues from the returned position."""
    from adapter_reader import poly_string_file_load
    data = b"x : { 1, 2, 3 }\nyy : { 4, 5 }\nz : { 6, 7}\n"
    rows, pos = poly_string_file_load(data, 2, 0, "u64")
    assert rows == [[1, 2, 3], [4, 5]]
    rows, pos = poly_string_file_load(data, 1, pos, "u64")
    assert rows == [[6]]                       # "7}" is the closing token: the glued value is lost
    assert poly_string_file_load(data, 1, pos, "u64")[0] == []
