#!/usr/bin/env python3
"""Generate tests/golden/ fixtures by driving the COMPILED REFERENCE (oracle/_ref/libse_ref.so,
built from /root/reference/device/lib with -O3 -fno-strict-aliasing by oracle/Makefile).

Run in the build container only:   python tests/golden/make_golden.py
Outputs (committed):  golden_c1.npz  golden_digests.json  ref_kats.json

Fixtures are data only: inputs (or the recipe in tests/vectors.py that makes them) and the
reference's outputs / digests.
"""
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import vectors as V  # noqa: E402
from oracle import pyoracle  # noqa: E402
from oracle.pyoracle import Reference  # noqa: E402

SEED_A = hashlib.shake_256(b"golden-share").digest(64)
SEED_B = hashlib.shake_256(b"golden-secret").digest(64)
SEED_PK = hashlib.shake_256(b"golden-pk").digest(64)
SEED_EP = hashlib.shake_256(b"golden-ep").digest(64)


def ends(a, k=8):
    a = np.asarray(a).ravel()
    return [int(x) for x in a[:k]] + [int(x) for x in a[-k:]]


def ntt_inputs(n, q, rng):
    delta = np.zeros(n, dtype=np.uint32)
    delta[1] = 1
    return {
        "delta1": delta,
        "ones": np.ones(n, dtype=np.uint32),
        "ramp": (np.arange(n, dtype=np.uint64) % q).astype(np.uint32),
        "qm1": np.full(n, q - 1, dtype=np.uint32),
        "random": rng.integers(0, q, n, dtype=np.uint64).astype(np.uint32),
    }


def api_digest(n, nprimes, asym, sk, pk0, pk1, q):
    """se_setup + se_encrypt_seeded through the real API in a scratch CWD."""
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.makedirs(os.path.join(td, "adapter_output_data"))
        sk.tofile(os.path.join(td, "adapter_output_data", f"sk_{n}.dat"))
        if asym:
            for j in range(nprimes):
                pk0[j].tofile(os.path.join(td, "adapter_output_data", f"pk0_ntt_{n}_{q[j]}.dat"))
                pk1[j].tofile(os.path.join(td, "adapter_output_data", f"pk1_ntt_{n}_{q[j]}.dat"))
        os.chdir(td)
        try:
            got, ncalls, out = Reference.api_encrypt(n, nprimes, asym, V.survey_values(n),
                                                     V.SURVEY_SHARE_SEED, V.SURVEY_SEED)
        finally:
            os.chdir(cwd)
    assert got == 8 * n * nprimes and ncalls == 2 * nprimes
    return "%016x" % pyoracle.fnv1a64(out.tobytes())


def api_print(n, nprimes, sk):
    """se_encrypt_seeded(print = true) in a scratch CWD: the lines the reference prints per prime."""
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.makedirs(os.path.join(td, "adapter_output_data"))
        sk.tofile(os.path.join(td, "adapter_output_data", f"sk_{n}.dat"))
        os.chdir(td)
        try:
            return Reference.api_print_lines(n, nprimes, False, V.survey_values(n), V.SURVEY_SHARE_SEED,
                                             V.SURVEY_SEED, os.path.join(td, "out.txt"))
        finally:
            os.chdir(cwd)


def main():
    pyoracle.build(ref=True)
    assert pyoracle.ref_available(), "needs /root/reference to build oracle/_ref"
    digests = {"generator": "tests/golden/make_golden.py", "shapes": {}}

    # ---- T8: twiddle-table digests of this host's libm (glibc 2.35) -------------------------
    from oracle.pyoracle import Oracle
    tw = {}
    for n in (1024, 2048, 4096, 8192, 16384):
        tw[str(n)] = hashlib.sha256(Oracle(n, 1).twiddles().astype("<f8").tobytes()).hexdigest()
    digests["ifft_twiddle_sha256"] = tw

    # ---- G1: PRNG blocks ----------------------------------------------------------------------
    digests["prng"] = [
        {"seed": "golden-share", "ctr": c, "len": ln,
         "sha256": hashlib.sha256(Reference.prng_block(SEED_A, c, ln)).hexdigest(),
         "head": Reference.prng_block(SEED_A, c, ln)[:16].hex()}
        for c, ln in [(0, 96), (7, 4), (2 ** 40 + 3, 1), (11, 16384), (12, 65536)]
    ]

    small = {}
    for (n, nprimes) in V.ALL_SHAPES:
        R = Reference(n, nprimes)
        sk = V.secret_key(n)
        R.set_sk(sk)
        q, lo, hi = R.moduli()
        rng = np.random.default_rng(1000 + n)
        d = {"q": [int(x) for x in q], "scale": R.scale()}
        d["index_map_sha256"] = V.sha256_hex(R.index_map())
        d["index_map_ends"] = ends(R.index_map())

        # G3 encode, nine patterns + survey + bench vector
        enc = {}
        for t in range(9):
            ok, m = R.encode(V.pattern_values(t, n))
            assert ok
            enc[f"pattern{t}"] = {"sha256": V.sha256_hex(m), "ends": ends(m)}
            if n == 1024:
                small[f"encode_pattern{t}"] = m
        ok, m = R.encode(V.bench_values(1, n)[0])
        enc["bench0"] = {"sha256": V.sha256_hex(m), "ends": ends(m)}
        big = np.full(n // 2, 3.0e38, dtype=np.float32)
        ok_big, _ = R.encode(big)
        enc["overflow_3e38_ok"] = bool(ok_big)
        # non-finite / edge-magnitude values: where ckks_encode_base gives up (n: nowhere) and what its
        # in-place loop had converted by then; accepted plaintexts (NaN -> INT64_MIN) also end to end
        nf = []
        for c in range(V.NONFINITE_CASES):
            vals = V.nonfinite_values(c, n)
            idx, m = R.encode_ex(vals)
            assert idx >= 0
            ent = {"fail_index": idx, "prefix_sha256": V.sha256_hex(m[:idx]),
                   "int64_min_count": int((m[:idx] == -2 ** 63).sum())}
            if idx == n:
                r = R.encrypt_sym(vals, SEED_A, SEED_B)
                assert r["ok"]
                ent["c0_sha256"], ent["pte_sha256"] = V.sha256_hex(r["c0"]), V.sha256_hex(r["pte"])
            nf.append(ent)
        enc["nonfinite"] = nf
        d["encode"] = enc

        # G4 samplers
        smp = {}
        ctr = 0
        for j in range(nprimes):
            a, ctr2 = R.sample_uniform(j, SEED_A, ctr)
            smp[f"uniform_p{j}"] = {"ctr_in": ctr, "ctr_out": ctr2, "sha256": V.sha256_hex(a),
                                    "ends": ends(a)}
            if n == 1024:
                small["uniform_p0"] = a
            ctr = ctr2
        u, c2 = R.sample_ternary_small(SEED_B, 0)
        smp["ternary"] = {"ctr_out": c2, "sha256": V.sha256_hex(u), "ends": ends(u)}
        e, c3 = R.cbd_int8(SEED_B, c2)
        smp["cbd_int8"] = {"ctr_in": c2, "ctr_out": c3, "sha256": V.sha256_hex(e), "ends": ends(e)}
        if n == 1024:
            small["ternary"], small["cbd_int8"] = u, e
        d["samplers"] = smp

        # G5 NTT KATs
        ntt = {}
        for j in range(nprimes):
            for name, x in ntt_inputs(n, int(q[j]), rng).items():
                y = R.ntt(x, j)
                ntt[f"{name}_p{j}"] = {"sha256": V.sha256_hex(y), "ends": ends(y)}
                if name == "random":
                    ntt[f"{name}_p{j}"]["in_sha256"] = V.sha256_hex(x)
                if n == 1024:
                    small[f"ntt_{name}_in"], small[f"ntt_{name}_out"] = x, y
            ntt[f"roots_p{j}"] = {"sha256": V.sha256_hex(R.ntt_roots(j))}
        d["ntt"] = ntt
        d["ntt_random_seed"] = 1000 + n

        # reduce edge: negative multiples of q map to q (ckks_common.c:234)
        x = np.zeros(n, dtype=np.int64)
        x[:6] = [0, -int(q[0]), int(q[0]), -1, 1, -(2 ** 62)]
        d["reduce_edge"] = [int(v) for v in R.reduce_pte(x, 0)[:6]]

        # G6 symmetric end to end (survey inputs and golden seeds)
        for tag, vals, s1, s2 in [("survey", V.survey_values(n), V.SURVEY_SHARE_SEED, V.SURVEY_SEED),
                                  ("bench0", V.bench_values(1, n)[0], SEED_A, SEED_B)]:
            r = R.encrypt_sym(vals, s1, s2)
            assert r["ok"]
            d[f"sym_{tag}"] = {
                "end_ctr": r["end_ctr"],
                "c0_sha256": V.sha256_hex(r["c0"]), "c1_sha256": V.sha256_hex(r["c1"]),
                "c1_alias_sha256": V.sha256_hex(r["c1_alias"]), "pte_sha256": V.sha256_hex(r["pte"]),
                "ntt_s_sha256": V.sha256_hex(r["ntt_s"]),
                "c0_ends": [ends(r["c0"][j]) for j in range(nprimes)],
                "c1_ends": [ends(r["c1"][j]) for j in range(nprimes)],
            }
            if n == 1024 and tag == "survey":
                small["sym_c0"], small["sym_c1"], small["sym_pte"] = r["c0"], r["c1"], r["pte"]
                small["sym_c1_alias"] = r["c1_alias"]

        # verification side: the reference's own decrypt / intt / decode on one of its ciphertexts
        vv = V.pattern_values(4, n)                       # all 1.1 (fits every parameter set)
        r = R.encrypt_sym(vv, SEED_A, SEED_B)
        ver = {}
        for j in range(nprimes):
            dec = R.decrypt(r["c0"][j], r["c1"][j], r["ntt_s"][j], j)
            assert (dec == r["c1_alias"][j]).all()
            ptj = R.intt(dec, j)
            val = R.decode(ptj, j)
            assert np.abs(val - vv).max() < 0.1          # ckks_tests_common.c:132
            ver[f"p{j}"] = {"pt_sha256": V.sha256_hex(ptj), "values_sha256": V.sha256_hex(val),
                            "pt_ends": ends(ptj), "values_head": [float(x) for x in val[:4]]}
        d["verify_pattern4"] = ver

        # G7 asymmetric end to end; pk from the reference's own gen_pk
        pk0, pk1 = Reference.gen_pk(n, nprimes, sk, SEED_PK, SEED_EP)
        RA = Reference(n, nprimes, asym=True)
        r = RA.encrypt_asym(V.survey_values(n), V.SURVEY_SEED, pk0, pk1)
        assert r["ok"]
        d["asym_survey"] = {
            "end_ctr": r["end_ctr"], "pk0_sha256": V.sha256_hex(pk0), "pk1_sha256": V.sha256_hex(pk1),
            "c0_sha256": V.sha256_hex(r["c0"]), "c1_sha256": V.sha256_hex(r["c1"]),
            "pte_sha256": V.sha256_hex(r["pte"]), "u_sha256": V.sha256_hex(r["u"]),
            "e1_sha256": V.sha256_hex(r["e1"]),
            "c0_ends": [ends(r["c0"][j]) for j in range(nprimes)],
            "c1_ends": [ends(r["c1"][j]) for j in range(nprimes)],
        }
        if n == 1024:
            small["asym_c0"], small["asym_c1"] = r["c0"], r["c1"]
            small["pk0"], small["pk1"] = pk0, pk1

        # API-level callback stream (reproduces the c1-alias quirk in sym mode)
        d["api_fnv1a64_sym"] = api_digest(n, nprimes, False, sk, None, None, q)
        d["api_fnv1a64_asym"] = api_digest(n, nprimes, True, sk, pk0, pk1, q)
        if (n, nprimes) in ((1024, 1), (4096, 3)):
            d["api_print_sym"] = api_print(n, nprimes, sk)
        digests["shapes"][f"{n}x{nprimes}"] = d
        RA.close()
        R.close()
        print("done", n, nprimes, d["api_fnv1a64_sym"], d["api_fnv1a64_asym"])

    # ---- text format of util_print.h (print_poly_full / print_poly_flpt_full) -----------------
    poly = np.array([0, 1, 134012928, 1053818880, 4294967295, 7, 65536, 123456789], dtype=np.uint32)
    vals = np.array([0.0, 1.0, -2.1, 1.1, 0.005, -0.005, 25.5, -100.125, 3.14159, 1e6], dtype=np.float32)
    with tempfile.TemporaryDirectory() as td:
        f1, f2 = os.path.join(td, "p.txt"), os.path.join(td, "v.txt")
        Reference.print_text(f1, "c0", poly=poly)
        Reference.print_text(f2, "x", values=vals)
        digests["text_format"] = {"poly": [int(x) for x in poly], "values": [float(x) for x in vals],
                                  "poly_line": open(f1).read(), "values_line": open(f2).read()}

    np.savez_compressed(os.path.join(HERE, "golden_c1.npz"), **small)
    with open(os.path.join(HERE, "golden_digests.json"), "w") as f:
        json.dump(digests, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
