"""Lane-level model of the wave-cooperative Keccak round (seal-embedded_amd/csrc/kernels/keccak.cuh, WaveKeccak /
wave_keccak_round) -- CPU only.

The round is cross-lane choreography: DPP row shifts with bank / row masks, v_permlane16_swap / v_permlane32_swap on a
register pair, ds_bpermute.  This file models those primitives on 64-element arrays exactly as the ISA documents them,
replays the round the way the header writes it, and checks it against a plain Keccak-f[1600] (itself pinned to hashlib)
on random states -- so that the layout argument of the header (which lanes are stale where, why lanes 40..47 may pick
up column parities, why row 3 stays zero) is executable without a GPU.  The GPU suite checks the real kernels; this
checks the REASONING, and fails if the header's constants drift from the model's.
"""
import hashlib
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "seal-embedded_amd", "csrc", "kernels", "keccak.cuh")
M32 = 0xFFFFFFFF
RHO = [0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14]   # [x + 5 y]


# ---- reference permutation (64-bit lanes, FIPS 202) -----------------------------------------------------------------
def rc(r):
    out, R = 0, 1
    step = lambda R: ((R << 1) ^ ((R >> 7) * 0x71)) & 0xFF
    for _ in range(7 * r):
        R = step(R)
    for j in range(7):
        if R & 1:
            out |= 1 << ((1 << j) - 1)
        R = step(R)
    return out


def rol64(v, r):
    r %= 64
    return ((v << r) | (v >> (64 - r))) & 0xFFFFFFFFFFFFFFFF if r else v


def keccak_round_ref(A, r):
    C = [A[x] ^ A[x + 5] ^ A[x + 10] ^ A[x + 15] ^ A[x + 20] for x in range(5)]
    D = [C[(x - 1) % 5] ^ rol64(C[(x + 1) % 5], 1) for x in range(5)]
    A = [A[i] ^ D[i % 5] for i in range(25)]
    B = [0] * 25
    for x in range(5):
        for y in range(5):
            B[y + 5 * ((2 * x + 3 * y) % 5)] = rol64(A[x + 5 * y], RHO[x + 5 * y])
    A = [B[i] ^ (~B[(i % 5 + 1) % 5 + 5 * (i // 5)] & B[(i % 5 + 2) % 5 + 5 * (i // 5)] & 0xFFFFFFFFFFFFFFFF) for i in range(25)]
    A[0] ^= rc(r)
    return A


def keccak_f_ref(A):
    for r in range(24):
        A = keccak_round_ref(A, r)
    return A


def test_reference_permutation_is_keccak():
    # SHA3-256 of a short message: one absorb, one permutation
    msg = b"seal-embedded wave form"
    block = bytearray(136)
    block[:len(msg)] = msg
    block[len(msg)] ^= 0x06
    block[135] ^= 0x80
    A = [int.from_bytes(block[8 * i:8 * i + 8], "little") if i < 17 else 0 for i in range(25)]
    A = keccak_f_ref(A)
    out = b"".join(a.to_bytes(8, "little") for a in A)[:32]
    assert out == hashlib.sha3_256(msg).digest()


# ---- the cross-lane primitives ----------------------------------------------------------------------------------------
def dpp(src, ctrl, old=None, row_mask=0xF, bank_mask=0xF, bound_ctrl=False):
    """v_mov_b32_dpp: lane i of a 16-lane row reads lane i + n (row_shl:n), i - n (row_shr:n) or (i - n) mod 16
    (row_ror:n) of the SAME row.  A lane whose row / bank (4 lanes) is masked off keeps `old`; a lane whose source falls
    outside the row gets 0 with bound_ctrl, else keeps `old`."""
    kind, n = ctrl
    out = np.array(old if old is not None else np.zeros(64, dtype=np.uint64), dtype=np.uint64).copy()
    for i in range(64):
        row, pos = divmod(i, 16)
        if not (row_mask >> row) & 1 or not (bank_mask >> (pos // 4)) & 1:
            continue
        s = {"shl": pos + n, "shr": pos - n, "ror": (pos - n) % 16}[kind]
        if 0 <= s < 16:
            out[i] = src[16 * row + s]
        elif bound_ctrl:
            out[i] = 0
    return out


def rows(v):
    return [v[16 * r:16 * r + 16].copy() for r in range(4)]


def permlane16_swap(a, b):
    """odd rows of the first register <-> even rows of the second"""
    ra, rb = rows(a), rows(b)
    return np.concatenate([ra[0], rb[0], ra[2], rb[2]]), np.concatenate([ra[1], rb[1], ra[3], rb[3]])


def permlane32_swap(a, b):
    """upper half of the first register <-> lower half of the second"""
    return np.concatenate([a[:32], b[:32]]), np.concatenate([a[32:], b[32:]])


def bpermute(addr, v):
    return np.array([v[(int(a) // 4) % 64] for a in addr], dtype=np.uint64)


def alignbit(a, b, s):
    """({a, b} >> s)[31:0], s taken mod 32; element-wise"""
    s = np.asarray(s, dtype=np.uint64) & np.uint64(31)
    return (((a.astype(np.uint64) << np.uint64(32)) | b.astype(np.uint64)) >> s) & np.uint64(M32)


# ---- the wave form ------------------------------------------------------------------------------------------------------
class Wave:
    def __init__(self):
        lane = np.arange(64)
        g, y = lane & 7, lane >> 3
        x = (g + 4) % 5
        self.in_state = y < 5
        self.primary = self.in_state & (g >= 1) & (g <= 5)
        self.index = np.where(self.primary, x + 5 * y, -1)
        R = np.where(self.in_state, np.array(RHO)[np.minimum(x + 5 * y, 24)], 0)
        self.swap = (R >= 32) | (R == 0)
        self.sh = ((32 - (R & 31)) & 31).astype(np.uint64)
        xs, ys = (3 * y + x) % 5, x
        self.pi_addr = np.where(self.in_state, 4 * (8 * ys + xs + 1), 4 * np.where(lane < 48, 63, lane))
        self.iota = (lane == 1)
        self.lo = np.zeros(64, dtype=np.uint64)
        self.hi = np.zeros(64, dtype=np.uint64)

    def load(self, A):
        """state lanes onto primary AND copy lanes (what wave_prng_absorb / pi leave)"""
        for i in range(64):
            g, y = i & 7, i >> 3
            if y < 5:
                v = A[(g + 4) % 5 + 5 * y]
                self.lo[i], self.hi[i] = v & M32, v >> 32
            else:
                self.lo[i] = self.hi[i] = 0

    def state(self):
        A = [0] * 25
        for i in range(64):
            if self.index[i] >= 0:
                A[self.index[i]] = int(self.lo[i]) | (int(self.hi[i]) << 32)
        return A

    def round(self, r, stale_copies=False):
        lo, hi = self.lo, self.hi
        if stale_copies:   # what chi leaves in lanes g = 6, 7: anything -- the round must not read them
            for i in range(40):
                if (i & 7) >= 6:
                    lo[i], hi[i] = 0xDEADBEEF, 0x0BADF00D
        # theta: column parity of both halves (wave_column_parity2)
        tl = lo ^ dpp(lo, ("ror", 8), bound_ctrl=True)
        th = hi ^ dpp(hi, ("ror", 8), bound_ctrl=True)
        x, y = permlane16_swap(tl, th)
        z = x ^ y
        x, y = permlane32_swap(z, z)
        w = x ^ y
        clo, chi_ = permlane16_swap(w, w)
        # D and the apply: the asm block of wave_keccak_round
        elo, ehi = alignbit(clo, chi_, 31), alignbit(chi_, clo, 31)
        glo = dpp(elo, ("shr", 3), bound_ctrl=True) ^ clo
        ghi = dpp(ehi, ("shr", 3), bound_ctrl=True) ^ chi_
        glo = np.where(self._enabled(bank_mask=0x5) & self._valid(("shl", 2)), dpp(elo, ("shl", 2)) ^ clo, glo)
        ghi = np.where(self._enabled(bank_mask=0x5) & self._valid(("shl", 2)), dpp(ehi, ("shl", 2)) ^ chi_, ghi)
        en = self._enabled(row_mask=0x7) & self._valid(("shr", 1))
        lo = np.where(en, dpp(glo, ("shr", 1)) ^ lo, lo)
        hi = np.where(en, dpp(ghi, ("shr", 1)) ^ hi, hi)
        # rho
        a, b = np.where(self.swap, hi, lo), np.where(self.swap, lo, hi)
        rlo, rhi = alignbit(a, b, self.sh), alignbit(b, a, self.sh)
        # pi
        blo, bhi = bpermute(self.pi_addr, rlo), bpermute(self.pi_addr, rhi)
        # chi, iota
        chi3 = lambda p, q, s: (p ^ (~q & s)) & np.uint64(M32)
        lo = chi3(blo, dpp(blo, ("shl", 1), bound_ctrl=True), dpp(blo, ("shl", 2), bound_ctrl=True))
        hi = chi3(bhi, dpp(bhi, ("shl", 1), bound_ctrl=True), dpp(bhi, ("shl", 2), bound_ctrl=True))
        k = rc(r)
        lo = np.where(self.iota, lo ^ np.uint64(k & M32), lo)
        hi = np.where(self.iota, hi ^ np.uint64(k >> 32), hi)
        self.lo, self.hi = lo, hi

    @staticmethod
    def _enabled(row_mask=0xF, bank_mask=0xF):
        i = np.arange(64)
        return (((row_mask >> (i // 16)) & 1) == 1) & (((bank_mask >> ((i % 16) // 4)) & 1) == 1)

    @staticmethod
    def _valid(ctrl):
        kind, n = ctrl
        pos = np.arange(64) % 16
        s = pos + n if kind == "shl" else pos - n
        return (s >= 0) & (s < 16)


def test_wave_round_is_the_keccak_round_and_keeps_its_lane_invariants():
    rng = np.random.default_rng(20260930)
    for trial in range(6):
        A = [int(v) for v in rng.integers(0, 1 << 64, 25, dtype=np.uint64)]
        w = Wave()
        w.load(A)
        for r in range(24):
            w.round(r, stale_copies=(r > 0))       # the copies right of column 4 are garbage after chi
            A = keccak_round_ref(A, r)
            assert w.state() == A, (trial, r)
            # lanes 40..63 hold zeros again after pi + chi (lanes 40..47 pull lane 63, row 3 is never written by theta)
            assert not w.lo[40:].any() and not w.hi[40:].any()
            # the left-hand copy (g = 0, column 4) is valid again: theta of column 0 reads it
            for y in range(5):
                assert int(w.lo[8 * y]) | (int(w.hi[8 * y]) << 32) == A[4 + 5 * y]


def test_model_constants_are_the_headers():
    src = open(HDR).read()
    body = src[src.index("kKeccakRho[25] = {"):]
    rho = [int(v) for v in re.findall(r"\d+", body[:body.index("};")].split("=", 1)[1])]
    assert rho == RHO
    for needle in ("const int x = (g + 4) % 5;",
                   "k.sh   = (uint32_t)((32 - (R & 31)) & 31);",
                   "const int xs = (3 * y + x) % 5, ys = x;",
                   "(uint32_t)(4 * (8 * ys + xs + 1)) : (uint32_t)(4 * (lane < 48 ? 63 : lane))",
                   "k.iota  = (lane == 1)",
                   "row_shr:3 row_mask:0xf bank_mask:0xf bound_ctrl:1",
                   "row_shl:2 row_mask:0xf bank_mask:0x5",
                   "row_shr:1 row_mask:0x7 bank_mask:0xf",
                   "dpp_row<0x128>(lo)",                                  # row_ror:8
                   "chi3(blo, dpp_row<0x101>(blo), dpp_row<0x102>(blo))",  # row_shl:1, row_shl:2
                   "__builtin_amdgcn_permlane16_swap(tl, th, false, false)",
                   "__builtin_amdgcn_permlane32_swap(z, z, false, false)",
                   "__builtin_amdgcn_permlane16_swap(w, w, false, false)"):
        assert needle in src, needle
