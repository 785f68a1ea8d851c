"""CPU checks of the generated phase-synchronised Keccak (seal-embedded_amd/csrc/kernels/keccak_sync.cuh):
the committed header is exactly what tools/keccak_sched.py --product writes, and the inline-asm text it holds --
interpreted here instruction by instruction (v_bitop3 / v_alignbit / v_xor / v_mov and the scalar loop control of the
block) -- computes SHAKE256(seed || le64(ctr)) for random messages (hashlib; the PRNG of rng.h:78-91).  The GPU suite
checks the same kernels against the oracle; this pins the generator without a GPU."""
import hashlib
import os
import re
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
HEADER = os.path.join(ROOT, "seal-embedded_amd", "csrc", "kernels", "keccak_sync.cuh")

RC64 = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B,
        0x0000000080000001, 0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088,
        0x0000000080008009, 0x000000008000000A, 0x000000008000808B, 0x800000000000008B, 0x8000000000008089,
        0x8000000000008003, 0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
        0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
RC32 = [w for r in RC64 for w in (r & 0xFFFFFFFF, r >> 32)]     # the table the kernels pass: kKeccakRC[24][2]
M32 = 0xFFFFFFFF


def asm_of(text, func):
    m = re.search(r"void " + func + r"\(.*?asm volatile\(\"(.*?)\"\s*\n\s*:", text, re.S)
    assert m, func
    return m.group(1).split("\\n\\t")


def run_asm(lines, vin):
    """Interpret one asm block: vin = {vgpr index: value}; returns the VGPR file.  %[rc] is table index 0."""
    v, s = dict(vin), {}
    scc = 0
    barriers = 0
    labels = {ln[:-1]: i for i, ln in enumerate(lines) if re.fullmatch(r"\d+:", ln)}

    def src(tok):
        tok = tok.strip()
        if tok.startswith("v"):
            return v[int(tok[1:])]
        if tok.startswith("s"):
            return s[int(tok[1:])]
        return int(tok, 0) & M32

    pc = 0
    while pc < len(lines):
        ln = lines[pc].strip()
        pc += 1
        if re.fullmatch(r"\d+:", ln):
            continue
        op, _, rest = ln.partition(" ")
        a = [t.strip() for t in rest.split(",")]
        if op == "v_mov_b32":
            v[int(a[0][1:])] = src(a[1])
        elif op == "v_xor_b32":
            v[int(a[0][1:])] = src(a[1]) ^ src(a[2])
        elif op == "v_alignbit_b32":
            hi, lo, sh = src(a[1]), src(a[2]), int(a[3], 0) & 31
            v[int(a[0][1:])] = (((hi << 32) | lo) >> sh) & M32
        elif op == "v_bitop3_b32":
            last, lut = a[3].split()
            lut = int(lut.split(":")[1], 0)
            x, y, z = src(a[1]), src(a[2]), src(last)
            r = 0
            for idx in range(8):
                if (lut >> idx) & 1:
                    r |= (x if idx & 4 else ~x) & (y if idx & 2 else ~y) & (z if idx & 1 else ~z)
            v[int(a[0][1:])] = r & M32
        elif op in ("s_load_dwordx2", "s_load_dwordx4"):
            lo = int(re.match(r"s\[(\d+):", a[0]).group(1))
            base = 0 if a[1] == "%[rc]" else s[int(re.match(r"s\[(\d+):", a[1]).group(1))]
            assert base % 4 == 0
            for k in range(2 if op.endswith("x2") else 4):
                s[lo + k] = RC32[base // 4 + k]
        elif op == "s_mov_b64":
            s[int(re.match(r"s\[(\d+):", a[0]).group(1))] = 0          # a byte offset into the table
        elif op == "s_add_u32":
            s[int(a[0][1:])] = src(a[1]) + int(a[2], 0)
        elif op == "s_addc_u32":
            pass
        elif op == "s_movk_i32":
            s[int(a[0][1:])] = int(a[1], 0)
        elif op == "s_sub_u32":
            s[int(a[0][1:])] = src(a[1]) - int(a[2], 0)
        elif op == "s_cmp_lg_u32":
            scc = 1 if src(a[0]) != int(a[1], 0) else 0
        elif op == "s_cbranch_scc1":
            if scc:
                pc = labels[a[0][:-1]]
        elif op == "s_barrier":
            barriers += 1
        elif op == "s_waitcnt":
            pass
        else:
            raise AssertionError("unexpected instruction in the generated block: " + ln)
    return v, barriers


def test_header_is_what_the_generator_writes():
    import keccak_sched
    text, total, peak = keccak_sched.product_header()
    assert open(HEADER).read() == text, "keccak_sync.cuh differs from tools/keccak_sched.py --product: regenerate it"
    assert peak <= 120        # v8 .. : leaves a 128-VGPR kernel room for its own values


def test_generated_blocks_compute_shake256():
    text = open(HEADER).read()
    base = int(re.search(r"state pinned to v(\d+)\.\.", text).group(1))
    rng = np.random.default_rng(20260930)
    for func, nwords in (("keccak_fresh96_sync", 24), ("keccak_fresh4_sync", 1)):
        lines = asm_of(text, func)
        for case in range(3):
            seed = rng.integers(0, 256, 64, dtype=np.uint8).tobytes()
            ctr = [0, 2**32 - 1, 2**63 + 12345][case]
            w = list(struct.unpack("<16I", seed)) + [ctr & M32, ctr >> 32]
            regs, barriers = run_asm(lines, {base + k: w[k] for k in range(18)})
            assert barriers == 96, (func, barriers)        # the contract of the header: 4 per round
            want = hashlib.shake_256(seed + struct.pack("<Q", ctr)).digest(4 * nwords)
            got = struct.pack("<%dI" % nwords, *[regs[base + k] for k in range(nwords)])
            assert got == want, (func, case)
