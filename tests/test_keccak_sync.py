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


# ------------------------------------------------------------------------------------------------------------------
# Guards against the toolchain.  The generated blocks hold 96 workgroup barriers each and rely on
# every live wave of the workgroup executing them the same number of times: a compiler that turned a guard in front
# of a block into an exec-masked region with a branch around it, clobbered a register the block writes, or duplicated
# / split the block would turn into a GPU hang or a silent desynchronisation, not into a wrong digest a parity test
# could see.  These tests look at (a) the constraint lists of the header and (b) the ISA hipcc emits for the three
# kernels that call the blocks.
# ------------------------------------------------------------------------------------------------------------------
SYNC_KERNELS = ("k_sample_cbd", "k_candidates", "k_sample_ternary_window")


def _asm_parts(text, func):
    m = re.search(r"void " + func + r"\(.*?asm volatile\(\"(.*?)\"\s*\n\s*:(.*?)\n\s*:(.*?)\n\s*:(.*?)\);", text, re.S)
    assert m, func
    return m.group(1).split("\\n\\t"), m.group(2), m.group(3), m.group(4)


def test_constraint_lists_cover_every_register_the_blocks_write():
    """Every VGPR / SGPR an instruction of a block WRITES is either one of the pinned in/out operands ("+{vN}") or in
    the clobber list; every register it READS before writing it is a pinned operand (or the table pointer); scc is
    clobbered; nothing outside v8..v77 / s16..s30 is touched."""
    text = open(HEADER).read()
    for func in ("keccak_fresh96_sync", "keccak_fresh4_sync"):
        lines, outs, ins, clob = _asm_parts(text, func)
        pinned = {int(x) for x in re.findall(r"\"\+\{v(\d+)\}\"", outs)}          # in/out: readable from the start
        outonly = {int(x) for x in re.findall(r"\"=&\{v(\d+)\}\"", outs)}       # early-clobber outputs: write first
        assert not re.search(r"\"[=+]&?v\"", outs), "only pinned registers: the block names physical registers"
        clobv = {int(x) for x in re.findall(r"\"v(\d+)\"", clob)}
        clobs = {int(x) for x in re.findall(r"\"s(\d+)\"", clob)}
        assert "\"scc\"" in clob
        written_v, written_s = set(), set()

        def regs(tok):
            tok = tok.strip()
            m = re.fullmatch(r"([vs])(\d+)", tok)
            if m:
                return m.group(1), [int(m.group(2))]
            m = re.fullmatch(r"([vs])\[(\d+):(\d+)\]", tok)
            if m:
                return m.group(1), list(range(int(m.group(2)), int(m.group(3)) + 1))
            return None, []

        for ln in lines:
            ln = ln.strip()
            if re.fullmatch(r"\d+:", ln) or ln.startswith(("s_barrier", "s_waitcnt", "s_cbranch", "s_cmp")):
                continue
            op, _, rest = ln.partition(" ")
            toks = [t.split()[0] if t.strip() else t for t in rest.split(",")]
            kind, dst = regs(toks[0])
            assert kind, ln
            for t in toks[1:]:
                k, src = regs(t)
                for r in src:
                    if k == "v":
                        assert r in pinned or r in written_v, (func, "reads v%d before writing it" % r, ln)
                    elif k == "s":
                        assert r in written_s, (func, "reads s%d before writing it" % r, ln)
            for r in dst:
                if kind == "v":
                    assert r in pinned or r in outonly or r in clobv, \
                        (func, "v%d written but neither pinned nor clobbered" % r, ln)
                    written_v.add(r)
                else:
                    assert r in clobs, (func, "s%d written but not clobbered" % r, ln)
                    written_s.add(r)
        assert min(written_v | pinned) >= 8 and max(written_v | pinned) <= 77, func
        assert written_s and min(written_s) >= 16 and max(written_s) <= 30, func
        assert outonly <= written_v, (func, "an output register the block never writes")


_ISA_CACHE = {}


def _samplers_isa():
    if "isa" not in _ISA_CACHE:
        import subprocess
        import tempfile
        csrc = os.path.join(ROOT, "seal-embedded_amd", "csrc")
        out = os.path.join(tempfile.mkdtemp(prefix="seamd_isa_"), "samplers.s")
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", "-I" + csrc,
               "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
               os.path.join(csrc, "kernels", "samplers.hip"), "-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        _ISA_CACHE["isa"] = open(out).read().splitlines()
    return _ISA_CACHE["isa"]


def _kernel_body(isa, name):
    """Lines of the kernel whose mangled symbol contains `name`, from its label to its s_endpgm / .Lfunc_end."""
    start = next(i for i, ln in enumerate(isa) if re.match(r"^_ZN5seamd\d+" + name + r"E\w*:", ln))
    end = next(i for i in range(start, len(isa)) if isa[i].startswith(".Lfunc_end"))
    return isa[start:end]


def test_compiled_sync_kernels_keep_the_barrier_contract():
    """ISA of k_sample_cbd, k_candidates and k_sample_ternary_window (hipcc -S --offload-arch=gfx950).  The contract of
    keccak_sync.cuh is that every wave of the workgroup that has not ENDED executes the block the same number of times
    (here: once).  What the compiler may legally do with the guards the kernels write in front of it --
    `if (!__any(live)) return; if (!live) return;` -- is (a) a scalar branch to the end, (b) ONE exec-masked region with
    an s_cbranch_execz to the end (what hipcc 7.2 emits: it folds the two tests into one s_and_saveexec), or (c) no
    branch at all, the wave running the block with an empty mask.  All three keep the contract.  What would break it:
    a path AROUND the block that does not end the wave (it would meet other barriers out of step), the block
    duplicated or split, or a branch back over it.  So:
      * the generated block appears exactly ONCE per kernel (one ;;#ASMSTART ... ;;#ASMEND region with barriers), with
        the header's static barrier count (4 per round-loop body + the prologue / epilogue rounds); the dynamic count,
        96, is the interpreter test's;
      * every branch in front of the block whose target lies behind the block's start targets code behind its END from
        which s_endpgm is reached without any s_barrier and without any further branch;
      * nothing behind the block branches back to or over it (k_sample_ternary_window's own barriers all lie behind);
      * no scratch: a spill in a kernel whose registers the block pins would have to go around v8..v77."""
    isa = _samplers_isa()
    header = open(HEADER).read()
    static_barriers = {f: sum(1 for ln in _asm_parts(header, f)[0] if ln.strip() == "s_barrier")
                       for f in ("keccak_fresh96_sync", "keccak_fresh4_sync")}
    which = {"k_sample_cbd": "keccak_fresh96_sync", "k_candidates": "keccak_fresh4_sync",
             "k_sample_ternary_window": "keccak_fresh96_sync"}
    branch_re = re.compile(r"\s+(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)")
    for name in SYNC_KERNELS:
        body = _kernel_body(isa, name)
        app = [i for i, ln in enumerate(body) if ln.strip().startswith(";;#ASMSTART")]
        noapp = [i for i, ln in enumerate(body) if ln.strip().startswith(";;#ASMEND")]
        assert len(app) == len(noapp), name
        # the block is the one asm region that holds barriers (other regions: empty opaque-value asm)
        blocks = [(a, b) for a, b in zip(app, noapp) if any(ln.strip().startswith("s_barrier") for ln in body[a:b])]
        assert len(blocks) == 1, (name, "the generated block must appear exactly once", len(blocks))
        a, b = blocks[0]
        assert sum(1 for ln in body[a:b] if ln.strip().startswith("s_barrier")) == static_barriers[which[name]], name
        assert not any(ln.strip().startswith("s_barrier") for ln in body[:a]), (name, "barrier in front of the block")
        labels = {ln.split(":")[0]: i for i, ln in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", ln)}
        skips = 0
        for i, ln in enumerate(body[:a]):
            m = branch_re.match(ln)
            if not m:
                continue
            target = labels[m.group(2)]
            if target <= a:
                continue                       # stays in front of the block
            assert target >= b, (name, "branch into the block", ln.strip())
            skips += 1
            j = target
            while j < len(body) and "s_endpgm" not in body[j]:
                assert not body[j].strip().startswith("s_barrier") and not branch_re.match(body[j]), \
                    (name, "a wave that goes around the block must end; found", body[j].strip())
                j += 1
            assert j < len(body), name
        assert skips <= 2, (name, skips)       # the whole-wave exit (and, at most, the masked-lane exit)
        for i, ln in enumerate(body[b:], b):
            m = branch_re.match(ln)
            if m:
                assert labels[m.group(2)] >= b, (name, "branch back over the synchronised block", ln.strip())
        text = "\n".join(body)
        assert "scratch_" not in text and "buffer_store" not in text and "buffer_load" not in text, name
    meta = "\n".join(isa)
    for name in SYNC_KERNELS:
        m = re.search(r"\.amdhsa_kernel _ZN5seamd\d+" + name + r"E\w*\n(?:.*\n){0,6}?\s+\.amdhsa_private_segment_fixed_size (\d+)", meta)
        assert m and int(m.group(1)) == 0, (name, m and m.group(1))


def test_wave_form_round_constants_are_the_table():
    """The wave-cooperative rounds take their constants from a constexpr LFSR (keccak.cuh, keccak_round_constant:
    FIPS 202 section 3.2.5) as template arguments instead of loading kKeccakRC; the header static_asserts four rows.
    Here: the same LFSR, transcribed, against ALL 24 rows of the table the other forms load (and against the
    constants hashlib's SHAKE256 implies: test_generated_blocks_compute_shake256 runs the table)."""
    src = open(os.path.join(ROOT, "seal-embedded_amd", "csrc", "kernels", "keccak.cuh")).read()
    body = src[src.index("kKeccakRC[24][2] = {"):]
    body = body[:body.index("};")]
    words = [int(w, 16) for w in re.findall(r"0x([0-9a-fA-F]{8})u", body)]
    assert len(words) == 48
    table = [words[2 * r] | (words[2 * r + 1] << 32) for r in range(24)]

    def lfsr_rc(r):
        rc, R = 0, 1
        step = lambda R: ((R << 1) ^ ((R >> 7) * 0x71)) & 0xFF
        for _ in range(7 * r):
            R = step(R)
        for j in range(7):
            if R & 1:
                rc |= 1 << ((1 << j) - 1)
            R = step(R)
        return rc

    assert [lfsr_rc(r) for r in range(24)] == table
    # the header's own transcription has the same shape (guards against an edit of one and not the other)
    fn = src[src.index("constexpr uint64_t keccak_round_constant(int r)"):]
    fn = fn[:fn.index("static_assert")]
    assert "((R << 1) ^ ((R >> 7) * 0x71u)) & 0xFFu" in fn and "1ull << ((1 << j) - 1)" in fn and "7 * r" in fn
    assert "wave_keccak_round<(uint32_t)rc, (uint32_t)(rc >> 32)>(k)" in src
