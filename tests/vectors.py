"""Deterministic synthetic inputs shared by the golden generator, the tests and bench.py.

Everything here is pure numpy/hashlib so the same bytes are produced in the build container
and on the GPU box.
"""
import hashlib
import random
import struct

import numpy as np

# (n, nprimes) shapes named in BASELINE.json / SURVEY.md section 8
C1 = (1024, 1)
C2 = (4096, 3)
C4 = (16384, 6)
ALL_SHAPES = [(1024, 1), (2048, 1), (4096, 3), (8192, 6), (16384, 6)]

MASK64 = (1 << 64) - 1


def splitmix64(x):
    """Vectorised splitmix64 finaliser over uint64 arrays (counter-based generator)."""
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def bench_values(B, n, seed=0xC0FFEE, first=0):
    """float32[B][n/2], i.i.d. (random byte) / -10 in [-25.5, 0] -- the distribution the
    reference bench uses (device/bench/bench_sym.c:92, bench_common.h:162-171).
    Element (b, i) depends only on (seed, first+b, i)."""
    out = np.empty((B, n // 2), dtype=np.float32)
    cols = np.arange(n // 2, dtype=np.uint64)[None, :]
    step = 2048                      # rows per block: bounds the temporary uint64 arrays to ~100 MB
    with np.errstate(over="ignore"):
        for lo in range(0, B, step):
            hi = min(B, lo + step)
            idx = np.arange(first + lo, first + hi, dtype=np.uint64)[:, None] * np.uint64(n // 2) + cols
            r = splitmix64(idx ^ np.uint64(seed))
            out[lo:hi] = (r >> np.uint64(56)).astype(np.float32) / np.float32(-10.0)
    return out


def derive_seeds(label, B, first=0):
    """uint8[B][64]: seed[b] = SHAKE256(label || le64(b))[0:64]."""
    out = np.empty((B, 64), dtype=np.uint8)
    lab = label.encode()
    for b in range(B):
        out[b] = np.frombuffer(hashlib.shake_256(lab + struct.pack("<Q", first + b)).digest(64),
                               dtype=np.uint8)
    return out


def bench_seeds(B, first=0):
    return derive_seeds("se-bench-share-seed", B, first), derive_seeds("se-bench-seed", B, first)


def secret_key(n, seed=1):
    """2-bit packed ternary secret key in the sk_<n>.dat format (adapter/fileops.cpp:58-75):
    MSB-first within each byte, codes 0/1/2 = -1/0/+1."""
    rng = random.Random(seed)
    codes = [rng.randrange(3) for _ in range(n)]
    b = bytearray(n // 4)
    for i, c in enumerate(codes):
        b[i // 4] |= c << (6 - 2 * (i % 4))
    return np.frombuffer(bytes(b), dtype=np.uint8).copy()


def survey_values(n):
    """The input used for the SURVEY 8(c) FNV digests."""
    i = np.arange(n // 2, dtype=np.uint64)
    with np.errstate(over="ignore"):
        v = ((i * np.uint64(2654435761)) % np.uint64(100000)).astype(np.float64) / 1000 - 50
    return v.astype(np.float32)


SURVEY_SHARE_SEED = bytes(range(64))
SURVEY_SEED = bytes(255 - k for k in range(64))


def pattern_values(testnum, n, seed=7):
    """The nine input patterns of device/test/ckks_tests_common.c:25-57 (patterns 7 and 8 draw
    their random nibbles/bytes from a seeded numpy generator instead of getrandom)."""
    vlen = n // 2
    v = np.zeros(vlen, dtype=np.float32)
    rng = np.random.default_rng(seed + testnum)
    if testnum == 0:
        v[0] = 1
    elif testnum == 1:
        v[0] = 2
    elif testnum == 2:
        v[:] = 1
    elif testnum == 3:
        v[:] = 2
    elif testnum == 4:
        v[:] = np.float32(1.1)
    elif testnum == 5:
        v[:] = np.float32(-2.1)
    elif testnum == 6:
        v[0::2] = 0
        v[1::2] = 1
    elif testnum == 7:
        v[:] = (rng.integers(0, 16, vlen).astype(np.float64) / -100.0).astype(np.float32)
    else:
        v[:] = (rng.integers(0, 256, vlen).astype(np.float64) / -10.0).astype(np.float32)
    return v


# Non-finite / edge-magnitude plaintext values.  The reference accepts them: a NaN coefficient passes
# the overflow test (fabs(NaN) > 2^63 is false, ckks_common.c:195) and is stored as (int64_t)NaN =
# INT64_MIN by the x86-64 build; an infinite coefficient is the `return false`; which coefficients
# are NaN and which infinite is decided by the Annex-G complex product of the IFFT (oracle/se_oracle.c,
# seo_cmul).  Cases 0-8 are fixed; 9-15 are seeded mixes of specials among ordinary values; 16-19 place
# a few +Inf and -Inf so that the FIRST coefficients come out NaN (accepted) and a later one infinite: the
# reference returns false at an index > 0 with a converted prefix behind it.
NONFINITE_CASES = 20
_SPECIALS = np.array([np.inf, -np.inf, np.nan, 3.4028235e38, -3.4028235e38, 1e-45, -1e-45, -0.0, 1e-39],
                     dtype=np.float32)


def nonfinite_values(case, n, seed=99):
    vlen = n // 2
    rng = np.random.default_rng(seed * 1000 + case * 17 + n)
    v = (rng.integers(0, 256, vlen).astype(np.float64) / -10.0).astype(np.float32)
    if case == 0:
        v[0] = np.nan
    elif case == 1:
        v[:] = np.nan
    elif case == 2:
        v[5] = np.inf
    elif case == 3:
        v[vlen - 1] = -np.inf
    elif case == 4:
        v[rng.random(vlen) < 0.25] = np.inf
        v[rng.random(vlen) < 0.25] = -np.inf
    elif case == 5:
        v[:] = -0.0
    elif case == 6:
        v[0::3], v[1::3], v[2::3] = np.float32(1e-45), np.float32(-1e-45), np.float32(1e-39)
    elif case == 7:
        v[:] = np.float32(3.4028235e38)
    elif case == 8:
        v[:] = 0
        v[vlen // 2] = np.float32(-3.4028235e38)
    elif case >= 16:
        if case < 18:
            v[:] = 0
        k = int(rng.integers(2, 5))
        pos = rng.choice(vlen, size=k, replace=False)
        v[pos] = np.where(rng.random(k) < 0.5, np.float32(np.inf), np.float32(-np.inf))
        v[pos[0]], v[pos[1]] = np.float32(np.inf), np.float32(-np.inf)
    else:
        k = int(rng.integers(1, 7))
        pool = _SPECIALS[rng.choice(len(_SPECIALS), size=int(rng.integers(1, 5)), replace=False)]
        v[rng.choice(vlen, size=k, replace=False)] = rng.choice(pool, size=k)
        if case % 3 == 0:
            v[rng.random(vlen) < 0.5] = rng.choice(pool)
    return v


def sha256_hex(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
