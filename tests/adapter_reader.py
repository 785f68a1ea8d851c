"""Restatement (test infrastructure, Python) of the SEAL-side reader of the adapter:

  poly_string_file_load   /root/reference/adapter/fileops.h:220-287
  ct_string_file_load     /root/reference/adapter/fileops.cpp:492-538
  the per-test sequence   /root/reference/adapter/adapter.cpp:94-127  (values line, then the ciphertext)

so that what the product's text writer emits (se_amd_write_ciphertext_text) can be
read back the way a machine with SEAL would read it, and compared with the SEAL `Ciphertext` data layout
(uint64 [component][prime][coeff], adapter/fileops.cpp:515-527) the packer produces.  SEAL itself (3.7.2) is
not buildable offline: this pins the adapter CONTRACT -- file position threading, brace scanning, token
rules, component order -- as far as this image allows; the arithmetic side is pinned by the device oracle.
"""
import numpy as np


def poly_string_file_load(data, ncomponents, pos=0, kind="u64"):
    """fileops.h:220-287 on the file's bytes: from `pos`, for each of `ncomponents` objects find the next '{',
    then read whitespace-separated tokens until one contains '}', strip commas, convert (stod / strtoull /
    strtoll); object k lands at vec[k * N ...] with N = its own token count.  Returns (rows, end position)."""
    conv = {"u64": lambda t: int(t, 10) & 0xFFFFFFFFFFFFFFFF, "i64": lambda t: int(t, 10), "f64": float}[kind]
    rows, i = [], pos
    n = len(data)
    while len(rows) < ncomponents and i < n:
        ch = data[i:i + 1]
        i += 1
        if ch != b"{":
            continue
        vals = []
        while i < n:
            # operator>>(string): skip whitespace, take the run of non-whitespace characters
            while i < n and data[i:i + 1].isspace():
                i += 1
            j = i
            while j < n and not data[j:j + 1].isspace():
                j += 1
            tok = data[i:j].decode()
            i = j
            if "}" in tok:
                break
            vals.append(conv(tok.replace(",", "")))
        rows.append(vals)
    return rows, i


def ct_string_file_load(data, n, nprimes, pos=0):
    """fileops.cpp:492-538: per prime two components (c0 line, c1 line) of n values; returns the SEAL
    Ciphertext data array (uint64, [component][prime][coeff] flattened) and the end position."""
    ct = np.zeros(2 * nprimes * n, dtype=np.uint64)
    for j in range(nprimes):
        rows, pos = poly_string_file_load(data, 2, pos, "u64")
        assert len(rows) == 2 and len(rows[0]) == n and len(rows[1]) == n, "a prime needs two n-value objects"
        ct[j * n:(j + 1) * n] = rows[0]
        ct[nprimes * n + j * n:nprimes * n + (j + 1) * n] = rows[1]
    return ct, pos


def read_tests(data, n, nprimes, ntests):
    """adapter.cpp:94-127: per test the values line (one object of n/2 doubles) and then the ciphertext,
    the file position threaded through."""
    out, pos = [], 0
    for _ in range(ntests):
        rows, pos = poly_string_file_load(data, 1, pos, "f64")
        ct, pos = ct_string_file_load(data, n, nprimes, pos)
        out.append((np.array(rows[0], dtype=np.float64), ct))
    return out
