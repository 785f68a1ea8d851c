#!/usr/bin/env python3
"""Schedule search for the lane-per-state Keccak-f[1600] round on gfx950 (round 4).

The chain kernels spend ~80 % of their instructions in the permutation and run it at ~3.7 issue cycles per
VALU instruction where the opcode mix alone would allow ~3.3.  This tool asks the hardware
which INSTRUCTION ORDER and REGISTER ASSIGNMENT it prefers: it builds the round as a data-flow graph (the same
190 operations per round as keccak.cuh's unfolded form: 70 v_bitop3, 58 v_alignbit, 62 v_xor), list-schedules it
under different priorities / minimum producer-consumer distances / rounds per loop body, colours the values onto
physical VGPRs (optionally steering register-file banks) and emits every variant as an inline-asm kernel of
tools/ubench7.hip, next to the compiler's own code for keccak.cuh.  ubench7 checks every variant against the
compiler's permutation and times it at 1, 2 and 4 waves per SIMD.

    python tools/keccak_sched.py            # writes tools/ubench7.hip
    hipcc --offload-arch=gfx950 -O3 tools/ubench7.hip -o tools/ubench7 && tools/ubench7

Replaces nothing in the reference (keccakf1600.c:51-316 is the function being computed); test tooling only.
"""
import os
import random
import sys

RHO = [0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14]


def pi_dst(i):
    x, y = i % 5, i // 5
    return y + 5 * ((2 * x + 3 * y) % 5)


class Ins:
    __slots__ = ("op", "dst", "srcs", "imm", "idx", "cls", "extra_deps", "phase")

    def __init__(self, op, dst, srcs, imm):
        self.op, self.dst, self.srcs, self.imm = op, dst, tuple(srcs), imm
        self.cls = {"bitop3": 1, "alignbit": 2, "xor": 0, "iota": 0}[op]
        self.extra_deps = []


def build(rounds, zero_in=(), need_out=None):
    """Data-flow graph of `rounds` consecutive rounds.  Values 0..49 are the incoming state (lane i half h = 2i+h);
    slots in `zero_in` are known to be zero (a freshly absorbed PRNG message: keccak.cuh, prng_absorb) and fold
    away; with `need_out` (slots) everything the other outgoing slots need is dropped.
    Returns (instructions in natural phase order, outgoing state values [50] (None = not produced), number of values)."""
    ins = []
    nv = [50]
    cur = [0]

    def emit(op, srcs, imm=None):
        nv[0] += 1
        ins.append(Ins(op, nv[0] - 1, srcs, imm))
        ins[-1].phase = cur[0]
        return nv[0] - 1

    def x2(a, b):
        if a is None:
            return b
        if b is None:
            return a
        return emit("xor", (a, b))

    def x5(vals):
        v = [t for t in vals if t is not None]
        assert v
        while len(v) > 3:
            v = [emit("bitop3", tuple(v[:3]), 0x96)] + v[3:]
        if len(v) == 3:
            return emit("bitop3", tuple(v), 0x96)
        if len(v) == 2:
            return emit("xor", tuple(v))
        return v[0]

    A = [[None if 2 * i + h in zero_in else 2 * i + h for h in (0, 1)] for i in range(25)]
    for r in range(rounds):
        cur[0] = 4 * r          # phases per round: parity (fast) | rol1 (slow) | D + theta (fast) | rho (slow) | chi + iota (fast, = next parity's run)
        C = [[x5([A[x + 5 * y][h] for y in range(5)]) for h in (0, 1)] for x in range(5)]
        cur[0] = 4 * r + 1
        R1 = [[emit("alignbit", (C[x][0], C[x][1]), 31), emit("alignbit", (C[x][1], C[x][0]), 31)] for x in range(5)]
        cur[0] = 4 * r + 2
        D = [[emit("xor", (C[(x + 4) % 5][h], R1[(x + 1) % 5][h])) for h in (0, 1)] for x in range(5)]
        B = [None] * 25
        for i in range(25):
            cur[0] = 4 * r + 2
            T = [x2(A[i][h], D[i % 5][h]) for h in (0, 1)]
            cur[0] = 4 * r + 3
            R = RHO[i]
            d = pi_dst(i)
            if R == 0:
                B[d] = T
            elif R < 32:
                B[d] = [emit("alignbit", (T[0], T[1]), 32 - R), emit("alignbit", (T[1], T[0]), 32 - R)]
            elif R == 32:
                B[d] = [T[1], T[0]]
            else:
                B[d] = [emit("alignbit", (T[1], T[0]), 64 - R), emit("alignbit", (T[0], T[1]), 64 - R)]
        cur[0] = 4 * r + 4
        An = [[None, None] for _ in range(25)]
        for y in range(0, 25, 5):
            for x in range(5):
                for h in (0, 1):
                    An[y + x][h] = emit("bitop3", (B[y + x][h], B[y + (x + 1) % 5][h], B[y + (x + 2) % 5][h]), 0xD2)
        for h in (0, 1):
            An[0][h] = emit("iota", (An[0][h],), (r, h))
        A = An
    outs = [A[i][h] for i in range(25) for h in (0, 1)]
    if need_out is not None:
        outs = [v if k in need_out else None for k, v in enumerate(outs)]
        live = set(v for v in outs if v is not None)
        kept = []
        for i in reversed(ins):
            if i.dst in live:
                kept.append(i)
                live.update(i.srcs)
        ins = kept[::-1]
    for k, i in enumerate(ins):
        i.idx = k
    return ins, outs, nv[0]


def schedule(ins, outs, nvals, prio="order", gap=1, seed=0, alt=False):
    """List scheduling.  prio: 'order' (natural phase order), 'height' (longest path first), 'random'.
    gap: an instruction is preferred only when every producer sits at least `gap` slots back.
    alt: prefer an instruction of another issue class than the previous one."""
    rnd = random.Random(seed)
    producer = {i.dst: i for i in ins}
    users = {}
    for i in ins:
        for s in i.srcs:
            users.setdefault(s, []).append(i)
    # WAR edges of the loop-carried registers: the value leaving in state slot k is written into the register the
    # incoming value k lives in, so every reader of incoming value k must come first
    for k, v in enumerate(outs):
        if v is not None:
            producer[v].extra_deps = [u for u in users.get(k, []) if u is not producer[v]]
    height = {}
    for i in reversed(ins):
        h = 0
        for u in users.get(i.dst, []):
            h = max(h, height[u.idx] + 1)
        height[i.idx] = h
    ndeps = {}
    rev = {}
    for i in ins:
        deps = set(producer[s].idx for s in i.srcs if s in producer) | set(d.idx for d in i.extra_deps)
        ndeps[i.idx] = len(deps)
        for d in deps:
            rev.setdefault(d, []).append(i.idx)
    ready = [i.idx for i in ins if ndeps[i.idx] == 0]
    pos = {}
    order = []
    last_cls = -1
    while ready:
        t = len(order)

        def dist_ok(k):
            for s in ins[k].srcs:
                if s in producer and t - pos[producer[s].idx] < gap:
                    return False
            return True

        cand = [k for k in ready if dist_ok(k)] or ready
        if alt:
            other = [k for k in cand if ins[k].cls != last_cls]
            cand = other or cand
        if prio == "order":
            k = min(cand)
        elif prio == "phased":               # whole phases: every run of one issue class as long as the data flow allows
            k = min(ready, key=lambda c: (ins[c].phase, c))
        elif prio == "height":
            k = max(cand, key=lambda c: (height[c], -c))
        elif prio == "random":
            k = rnd.choice(cand)
        elif prio == "hrand":
            k = max(cand, key=lambda c: (height[c] + rnd.random() * 3.0))
        else:
            raise ValueError(prio)
        ready.remove(k)
        pos[k] = t
        order.append(ins[k])
        last_cls = ins[k].cls
        for u in rev.get(k, []):
            ndeps[u] -= 1
            if ndeps[u] == 0:
                ready.append(u)
    assert len(order) == len(ins)
    return order


def allocate(order, outs, nvals, ntemps, banks=None, zero_in=()):
    """Interval colouring onto registers 0..49 (state; incoming value k and outgoing value of slot k share register k)
    and 50..50+ntemps-1.  banks: None, or 'spread' = prefer a destination bank different from the banks of the
    value's future co-operands (register file banks = register index mod 4)."""
    END = len(order) + 1
    defpos = {k: -1 for k in range(50)}
    last = {k: -1 for k in range(50)}
    for p, i in enumerate(order):
        defpos[i.dst] = p
        last.setdefault(i.dst, p)
        for s in i.srcs:
            last[s] = p
    out_slot = {v: k for k, v in enumerate(outs) if v is not None}
    for v in out_slot:
        last[v] = END
    nregs = 50 + ntemps
    occ = [[] for _ in range(nregs)]     # per register: list of (start, end) = (def, last use)
    reg = {}
    for k in range(50):
        if k in zero_in:
            continue                      # no incoming value: the register is free from the start
        reg[k] = k
        occ[k].append((-1, last[k]))
    for v, k in out_slot.items():
        assert k in zero_in or defpos[v] >= last[k], "loop-carried register still live"
        reg[v] = k
        occ[k].append((defpos[v], END))
    consumers = {}
    for i in order:
        for s in i.srcs:
            consumers.setdefault(s, []).append(i)

    def free(r, a, b):
        for (s, e) in occ[r]:
            if a < e and s < b:
                return False
        return True

    peak = 0
    for p, i in enumerate(order):
        v = i.dst
        if v in reg:
            continue
        a, b = p, last[v]
        cands = [r for r in range(nregs) if free(r, a, b)]
        if not cands:
            raise RuntimeError("out of registers")
        if banks == "spread":
            avoid = {}
            for c in consumers.get(v, []):
                for s in c.srcs:
                    if s != v and s in reg:
                        avoid[reg[s] % 4] = avoid.get(reg[s] % 4, 0) + 1
            for s in i.srcs:             # and this instruction's own sources (write port vs read ports)
                if s in reg:
                    avoid[reg[s] % 4] = avoid.get(reg[s] % 4, 0) + 0.5
            r = min(cands, key=lambda c: (avoid.get(c % 4, 0), c))
        elif banks == "same":            # adversarial: pile everything on the sources' banks
            want = [reg[s] % 4 for s in i.srcs if s in reg]
            r = min(cands, key=lambda c: (0 if (want and c % 4 == want[0]) else 1, c))
        else:
            r = cands[0]
        reg[v] = r
        occ[r].append((a, b))
        peak = max(peak, r + 1)
    return reg, peak


def emit_asm(order, reg, base, sreg_rc, subst=None, bar=0):
    """Text of the body: one instruction per line, physical registers v{base + r}; iota reads s{sreg_rc + 2 r + h}.
    An s_waitcnt lgkmcnt(0) precedes the first iota (the round constants arrive by s_load)."""
    out = []
    waited = False
    prev_slow = None
    for i in order:
        slow = i.op == "alignbit"
        # bar = 4: a workgroup barrier at every change of issue class; bar = 2: only around the long rho run
        if bar in (2, 4) and prev_slow is not None and slow != prev_slow and (bar == 4 or i.phase % 4 == 3 or (i.phase % 4 == 0 and prev_slow)):
            out.append("s_barrier")
        if bar == 1 and prev_slow is not None and slow and not prev_slow and i.phase % 4 == 3:
            out.append("s_barrier")          # once per round: in front of the rho run
        if bar == 9 and prev_slow is None:
            out.append("s_barrier")          # once per loop body
        prev_slow = slow
        d = f"v{base + reg[i.dst]}"
        s = [f"v{base + reg[x]}" for x in i.srcs]
        to = (subst or {}).get(i.op)
        if len(s) == 1:
            s = [s[0], s[0]]                  # iota: one vector source
        if to == "xor":                       # timing ablation (WRONG results): the same registers through v_xor
            out.append(f"v_xor_b32 {d}, {s[0]}, {s[1]}")
        elif to == "alignbit":
            out.append(f"v_alignbit_b32 {d}, {s[0]}, {s[1]}, 7")
        elif to == "bitop3":
            out.append(f"v_bitop3_b32 {d}, {s[0]}, {s[1]}, {s[-1]} bitop3:0x96")
        elif i.op == "bitop3":
            out.append(f"v_bitop3_b32 {d}, {s[0]}, {s[1]}, {s[2]} bitop3:{hex(i.imm)}")
        elif i.op == "alignbit":
            out.append(f"v_alignbit_b32 {d}, {s[0]}, {s[1]}, {i.imm}")
        elif i.op == "xor":
            out.append(f"v_xor_b32 {d}, {s[0]}, {s[1]}")
        else:
            if not waited:
                out.append("s_waitcnt lgkmcnt(0)")
                waited = True
            r, h = i.imm
            out.append(f"v_xor_b32 {d}, s{sreg_rc + 2 * r + h}, {s[0]}")
    return out


def stats(order):
    producer_pos = {}
    dep1 = 0
    for p, i in enumerate(order):
        for s in i.srcs:
            if producer_pos.get(s) == p - 1:
                dep1 += 1
                break
        producer_pos[i.dst] = p
    return dep1


VARIANTS = [
    # name, rounds per loop body, prio, gap, alt, seed, banks
    ("order_r2", 2, "order", 1, False, 0, None),
    ("order_r1", 1, "order", 1, False, 0, None),
    ("order_r2_g2", 2, "order", 2, False, 0, None),
    ("order_r2_g4", 2, "order", 4, False, 0, None),
    ("height_r2_g1", 2, "height", 1, False, 0, None),
    ("height_r2_g2", 2, "height", 2, False, 0, None),
    ("height_r2_g4", 2, "height", 4, False, 0, None),
    ("height_r2_g8", 2, "height", 8, False, 0, None),
    ("height_r1_g2", 1, "height", 2, False, 0, None),
    ("height_r4_g2", 4, "height", 2, False, 0, None),
    ("order_r2_alt", 2, "order", 2, True, 0, None),
    ("height_r2_alt", 2, "height", 2, True, 0, None),
    ("hrand_r2_a", 2, "hrand", 2, False, 1, None),
    ("hrand_r2_b", 2, "hrand", 2, False, 2, None),
    ("random_r2_a", 2, "random", 2, False, 3, None),
    ("random_r2_b", 2, "random", 1, False, 4, None),
    ("order_r2_spread", 2, "order", 2, False, 0, "spread"),
    ("height_r2_spread", 2, "height", 2, False, 0, "spread"),
    ("order_r2_samebank", 2, "order", 2, False, 0, "same"),
    ("height_r2_samebank", 2, "height", 2, False, 0, "same"),
    # timing ablations of order_r2_g2 (WRONG results by construction): which opcode class costs what IN THIS STREAM
    ("ABL_align_as_xor", 2, "order", 2, False, 0, None, {"alignbit": "xor"}),
    ("ABL_bitop_as_xor", 2, "order", 2, False, 0, None, {"bitop3": "xor"}),
    ("ABL_all_xor", 2, "order", 2, False, 0, None, {"bitop3": "xor", "alignbit": "xor"}),
    ("ABL_all_alignbit", 2, "order", 2, False, 0, None, {"bitop3": "alignbit", "xor": "alignbit", "iota": "alignbit"}),
    ("ABL_all_bitop3", 2, "order", 2, False, 0, None, {"xor": "bitop3", "alignbit": "bitop3"}),
    ("ABL_xor_as_bitop3", 2, "order", 2, False, 0, None, {"xor": "bitop3"}),
    # whole-phase order (runs of one issue class), without and with workgroup barriers at the class changes; 1024-thread
    # workgroups put 4 waves of ONE workgroup on every SIMD, so a barrier aligns the waves that share a SIMD
    ("phased_r2", 2, "phased", 1, False, 0, None, None, 0, 256),
    ("phased_r1", 1, "phased", 1, False, 0, None, None, 0, 256),
    ("order_r2_t1024", 2, "order", 2, False, 0, None, None, 0, 1024),
    ("phased_r2_t1024", 2, "phased", 1, False, 0, None, None, 0, 1024),
    ("phased_r2_bar4", 2, "phased", 1, False, 0, None, None, 4, 1024),
    ("phased_r2_bar2", 2, "phased", 1, False, 0, None, None, 2, 1024),
    ("phased_r1_bar4", 1, "phased", 1, False, 0, None, None, 4, 1024),
    ("order_r2_bar2", 2, "order", 2, False, 0, None, None, 2, 1024),
    ("phased_r2_bar4_t512", 2, "phased", 1, False, 0, None, None, 4, 512),
    ("phased_r2_bar2_t512", 2, "phased", 1, False, 0, None, None, 2, 512),
    ("phased_r2_bar1_t512", 2, "phased", 1, False, 0, None, None, 1, 512),
    ("phased_r2_bar9_t512", 2, "phased", 1, False, 0, None, None, 9, 512),
    ("phased_r4_bar9_t512", 4, "phased", 1, False, 0, None, None, 9, 512),
    ("order_r2_bar4_t512", 2, "order", 2, False, 0, None, None, 4, 512),
    ("phased_r2_bar2_t768", 2, "phased", 1, False, 0, None, None, 2, 768),
]

BASE = 8           # first physical VGPR of the state
NTEMPS = 70        # v58 .. v127: the kernel stays within 128 VGPRs (4 waves per SIMD)
SRC = 16           # s16.. : round constants of the loop body (2 per round)


def kernel_text(name, rounds, prio, gap, alt, seed, banks, subst=None, bar=0, threads=256):
    ins, outs, nvals = build(rounds)
    order = schedule(ins, outs, nvals, prio, gap, seed, alt)
    reg, peak = allocate(order, outs, nvals, NTEMPS, banks)
    body = emit_asm(order, reg, BASE, SRC, subst, bar)
    ndw = 2 * rounds
    assert ndw in (2, 4, 8)
    load = {2: "s_load_dwordx2 s[16:17], s[28:29], 0x0", 4: "s_load_dwordx4 s[16:19], s[28:29], 0x0",
            8: "s_load_dwordx8 s[16:23], s[28:29], 0x0"}[ndw]
    lines = ["s_mov_b32 s31, %[perms]", "2:", "s_mov_b64 s[28:29], %[rc]", f"s_movk_i32 s30, {24 // rounds}", "1:",
             load, f"s_add_u32 s28, s28, {4 * ndw}", "s_addc_u32 s29, s29, 0"] + body + [
             "s_sub_u32 s30, s30, 1", "s_cmp_lg_u32 s30, 0", "s_cbranch_scc1 1b",
             "s_sub_u32 s31, s31, 1", "s_cmp_lg_u32 s31, 0", "s_cbranch_scc1 2b"]
    asm = "\\n\\t".join(lines)
    ops = ", ".join(f'"+{{v{BASE + k}}}"(s[{k}])' for k in range(50))
    clob = ", ".join(f'"v{BASE + 50 + t}"' for t in range(peak - 50)) if peak > 50 else ""
    sclob = ", ".join(f'"s{n}"' for n in list(range(16, 24)) + [28, 29, 30, 31])
    info = f"{len(order)} instructions per body, {stats(order)} consume the result of the instruction right before them, {peak} VGPRs"
    return f'''// {name}: rounds/body {rounds}, priority {prio}, gap {gap}, alternate classes {alt}, banks {banks}: {info}
__global__ __launch_bounds__({threads}) void k_{name}(uint32_t* out, const uint32_t* rc, int perms)
{{
    uint32_t s[50];
    init_state(s);
    asm volatile("{asm}"
                 : {ops}
                 : [rc] "s"(rc), [perms] "s"(perms)
                 : {clob}{", " if clob else ""}{sclob}, "scc", "memory");
    store_state(out, s);
}}
''', info


PROD_BASE = 8          # the product forms pin the state to v8..v57, temporaries follow
PROD_TEMPS = 70


def _segment(rounds, zero_in=(), need_out=None, bar=4):
    ins, outs, nvals = build(rounds, zero_in, need_out)
    order = schedule(ins, outs, nvals, "phased", 1, 0, False)
    reg, peak = allocate(order, outs, nvals, PROD_TEMPS, None, zero_in)
    return emit_asm(order, reg, PROD_BASE, SRC, None, bar), peak, len(order)


def product_header(full=False):
    """seal-embedded_amd/csrc/kernels/keccak_sync.cuh: the permutation of a freshly absorbed PRNG message (keccak.cuh,
    prng_absorb) whose caller consumes the first 96 bytes, for kernels in which SEVERAL WAVES OF ONE WORKGROUP SHARE A
    SIMD: whole-phase instruction order and an s_barrier at every change of issue class."""
    present = set(range(19)) | {33}                  # seed (lanes 0..7), counter (lane 8), 0x1F (lane 9 lo), lane 16 hi
    zero_in = set(range(50)) - present
    seg0, p0, n0 = _segment(1, zero_in)
    segm, pm, nm = _segment(2)
    segl, pl, nl = _segment(1, (), set(range(24)))
    seg1, p1, n1 = _segment(1, (), {0})               # last round of a block whose caller consumes the first word only
    peak = max(p0, pm, pl, p1)
    B = PROD_BASE
    lines = [f"v_mov_b32 v{B + 18}, 0x1f", f"v_mov_b32 v{B + 33}, 0x80000000",
             "s_load_dwordx2 s[16:17], %[rc], 0x0"] + seg0 + [
             "s_mov_b64 s[28:29], %[rc]", "s_add_u32 s28, s28, 8", "s_addc_u32 s29, s29, 0", "s_movk_i32 s30, 11", "1:",
             "s_load_dwordx4 s[16:19], s[28:29], 0x0", "s_add_u32 s28, s28, 16", "s_addc_u32 s29, s29, 0"] + segm + [
             "s_sub_u32 s30, s30, 1", "s_cmp_lg_u32 s30, 0", "s_cbranch_scc1 1b",
             "s_load_dwordx2 s[16:17], s[28:29], 0x0"] + segl
    asm4 = "\\n\\t".join(lines[:len(lines) - len(segl)] + seg1)
    ops4 = ", ".join(f'"+{{v{B + k}}}"(w[{k}])' for k in range(18))
    clob4 = ", ".join(f'"v{B + k}"' for k in range(18, peak))
    total4 = n0 + 11 * nm + n1
    asm = "\\n\\t".join(lines)
    ops = ", ".join([f'"+{{v{B + k}}}"(w[{k}])' for k in range(18)] + [f'"=&{{v{B + k}}}"(w[{k}])' for k in range(18, 24)])
    clob = ", ".join(f'"v{B + k}"' for k in range(24, peak))
    sclob = ", ".join(f'"s{n}"' for n in (16, 17, 18, 19, 28, 29, 30))
    total = n0 + 11 * nm + nl
    full_lines = ["s_mov_b64 s[28:29], %[rc]", "s_movk_i32 s30, 12", "1:",
                  "s_load_dwordx4 s[16:19], s[28:29], 0x0", "s_add_u32 s28, s28, 16", "s_addc_u32 s29, s29, 0"] + segm + [
                  "s_sub_u32 s30, s30, 1", "s_cmp_lg_u32 s30, 0", "s_cbranch_scc1 1b"]
    asm_full = "\\n\\t".join(full_lines)
    ops_full = ", ".join(f'"+{{v{B + k}}}"(s[{k}])' for k in range(50))
    clob_full = ", ".join(f'"v{B + k}"' for k in range(50, pm))
    null_asm = "\\n\\t".join([f"s_movk_i32 s30, {4 * 24}", "1:", "s_barrier", "s_sub_u32 s30, s30, 1", "s_cmp_lg_u32 s30, 0",
                              "s_cbranch_scc1 1b"])
    # --full: also the state-in / state-out permutation (round 5: the paired chain kernel k_bulk_lane_sync squeezed with it,
    # experiments/r05_staged_lane_paired_chains.patch) and the barrier-only form (the lockstep sampler experiment of round
    # 4, profiles/r04_ab_lockstep.log); the product header carries neither
    extras = f'''// The full permutation, state in and out (lane i = s[2 i] (lo), s[2 i + 1] (hi)): the squeeze step of a sponge whose
// state stays in registers (12 iterations of the two-round body, {4 * 24} barriers, {12 * nm} VALU instructions).
__device__ __forceinline__ void keccak_f1600_sync(uint32_t (&s)[50], const uint32_t *rc)
{{
    asm volatile("{asm_full}"
                 : {ops_full}
                 : [rc] "s"(rc)
                 : {clob_full}, {sclob}, "scc");
}}

// For a wave of the workgroup that has no permutation to run while the others do: the same {4 * 24} barriers, no work.
__device__ __forceinline__ void keccak_null_sync()
{{
    asm volatile("{null_asm}" : : : "s30", "scc");
}}

''' if full else ""
    return f'''// keccak_sync.cuh -- GENERATED by tools/keccak_sched.py --product; do not edit.
//
// Keccak-f[1600] of a freshly absorbed PRNG message whose caller consumes the first 96 output bytes, as ONE inline-asm
// block with the state pinned to v{B}..v{B + 49}, for kernels in which several waves of one workgroup share a SIMD.
//
// Why (tools/keccak_sched.py, profiles/r04_ubench7_keccak_schedules.txt): gfx950 issues v_xor / v_bitop3 at ~2.1 / 2.8
// cycles per wave instruction only when TWO WAVES of a SIMD present such an instruction at the same time; next to a
// v_alignbit (4.3 cycles, the other third of the round) of another wave they take a full slot each, and waves running
// the same round unsynchronised settle out of phase (3.8-3.9 cycles per instruction whatever the instruction order or
// the register assignment).  Here every round runs in whole phases -- parities | rol1 | theta | rho | chi -- with an
// s_barrier at every change of issue class, so the waves of the workgroup that share a SIMD stay in phase: 2.96-3.2
// cycles per instruction, 11.5-11.8 G permutations/s chip-wide against 8.9-9.9 for the compiler's code.
// Same Boolean function as keccak.cuh (keccakf1600.c:51-316 of the reference); round 0 folds the known-zero lanes of
// the absorbed message, the last round only produces lanes 0..11.  {total} VALU instructions per permutation.
//
// CONTRACT: every wave of the workgroup that has not ended executes this block the same number of times (it contains
// {4 * 24} workgroup barriers); launch the kernel with at least 2 waves per SIMD and workgroup (>= 512 threads).  A path
// AROUND a call must be wave-uniform and must END the wave (`if (!__any(live)) return;`): a wave that went around the
// block and lived on would meet the workgroup's later barriers out of step.  Lanes of a partially active wave may be
// masked off, and a wave may even run the block with an empty mask (the barriers do not depend on exec): it runs the
// block once either way.  tests/test_keccak_sync.py checks the compiled ISA of every caller for exactly this.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace seamd {{

// in: w[0..15] = seed words, w[16], w[17] = counter (lo, hi); out: w[0..23] = the first 96 bytes of the block
__device__ __forceinline__ void keccak_fresh96_sync(uint32_t (&w)[24], const uint32_t *rc)
{{
    asm volatile("{asm}"
                 : {ops}
                 : [rc] "s"(rc)
                 : {clob}, {sclob}, "scc");
}}

// the same block for a caller that consumes its first 4 bytes (a redraw candidate, sample.c:54): in w[0..17] as above,
// out w[0]; the last round shrinks to what that word needs ({total4} VALU instructions per permutation)
__device__ __forceinline__ void keccak_fresh4_sync(uint32_t (&w)[18], const uint32_t *rc)
{{
    asm volatile("{asm4}"
                 : {ops4}
                 : [rc] "s"(rc)
                 : {clob4}, {sclob}, "scc");
}}

{extras}}}  // namespace seamd
''', total, peak


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    if "--product" in sys.argv:
        text, total, peak = product_header("--full" in sys.argv)
        path = os.path.join(here, "..", "seal-embedded_amd", "csrc", "kernels", "keccak_sync.cuh")
        with open(path, "w") as f:
            f.write(text)
        print(f"keccak_sync.cuh: {total} VALU instructions per permutation, registers v{PROD_BASE}..v{PROD_BASE + peak - 1}")
        return
    parts = ['''// GENERATED by tools/keccak_sched.py -- do not edit.  Schedule / register-assignment variants of the lane-per-state
// Keccak-f[1600] round as inline asm, checked against and timed beside the compiler's code for keccak.cuh.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <vector>
#include "../seal-embedded_amd/csrc/kernels/keccak.cuh"

__device__ __forceinline__ void init_state(uint32_t (&s)[50])
{
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;     // the same states whatever the workgroup size
    for (int i = 0; i < 50; i++) s[i] = (gid * 2654435761u) ^ (i * 40503u + (gid >> 8) * 977u);
}
__device__ __forceinline__ void store_state(uint32_t* out, const uint32_t (&s)[50])
{
    // every lane folds its state to 2 words; lane 0 of every block also dumps the full state
    uint32_t a = 0, b = 0;
    for (int i = 0; i < 50; i++) a ^= s[i] * (2 * i + 1), b += s[i] ^ (i << 7);
    uint32_t* o = out + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    o[0] = a, o[1] = b;
}
__global__ __launch_bounds__(256) void k_compiler(uint32_t* out, const uint32_t* rc, int perms)
{
    uint32_t s[50];
    init_state(s);
    seamd::KeccakState st;
    for (int i = 0; i < 25; i++) st.lo[i] = s[2 * i], st.hi[i] = s[2 * i + 1];
    for (int p = 0; p < perms; p++) seamd::keccak_f1600<false>(st);
    for (int i = 0; i < 25; i++) s[2 * i] = st.lo[i], s[2 * i + 1] = st.hi[i];
    store_state(out, s);
}
__global__ __launch_bounds__(256) void k_compiler_folded(uint32_t* out, const uint32_t* rc, int perms)
{
    uint32_t s[50];
    init_state(s);
    seamd::KeccakState st;
    for (int i = 0; i < 25; i++) st.lo[i] = s[2 * i], st.hi[i] = s[2 * i + 1];
    for (int p = 0; p < perms; p++) seamd::keccak_f1600<true>(st);
    for (int i = 0; i < 25; i++) s[2 * i] = st.lo[i], s[2 * i + 1] = st.hi[i];
    store_state(out, s);
}
''']
    names = []
    thr = {}
    for v in VARIANTS:
        try:
            text, info = kernel_text(*v)
        except RuntimeError as e:
            print(f"{v[0]:22s} skipped: {e}")
            continue
        parts.append(text)
        names.append(v[0])
        thr[v[0]] = v[9] if len(v) > 9 else 256
        print(f"{v[0]:22s} {info}")
    parts.append('''
static const uint32_t kRC[48] = {
    0x00000001u, 0x00000000u, 0x00008082u, 0x00000000u, 0x0000808au, 0x80000000u, 0x80008000u, 0x80000000u,
    0x0000808bu, 0x00000000u, 0x80000001u, 0x00000000u, 0x80008081u, 0x80000000u, 0x00008009u, 0x80000000u,
    0x0000008au, 0x00000000u, 0x00000088u, 0x00000000u, 0x80008009u, 0x00000000u, 0x8000000au, 0x00000000u,
    0x8000808bu, 0x00000000u, 0x0000008bu, 0x80000000u, 0x00008089u, 0x80000000u, 0x00008003u, 0x80000000u,
    0x00008002u, 0x80000000u, 0x00000080u, 0x80000000u, 0x0000800au, 0x00000000u, 0x8000000au, 0x80000000u,
    0x80008081u, 0x80000000u, 0x00008080u, 0x80000000u, 0x80000001u, 0x00000000u, 0x80008008u, 0x80000000u};

typedef void (*kern_t)(uint32_t*, const uint32_t*, int);
static std::vector<uint32_t> g_ref;

static void run(const char* name, kern_t kern, uint32_t* d, const uint32_t* rc, bool is_ref, int threads = 256)
{
    const int perms = 64;
    const bool ablation = strncmp(name, "ABL_", 4) == 0;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 100 << 10);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    // correctness: 3 permutations on 256 blocks against the compiler's code
    {
        const int blocks = 256 * 256 / threads;
        (void)hipMemset(d, 0, (size_t)65536 * 8);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, rc, 3);
        hipError_t e1 = hipGetLastError(), e2 = hipDeviceSynchronize();
        if (e1 != hipSuccess || e2 != hipSuccess) { printf("%-22s launch failed: %s / %s\\n", name, hipGetErrorString(e1), hipGetErrorString(e2)); return; }
        std::vector<uint32_t> h((size_t)65536 * 2);
        (void)hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        if (is_ref) g_ref = h;
        else if (ablation) {}
        else if (g_ref.size() != h.size() || memcmp(g_ref.data(), h.data(), h.size() * 4) != 0)
        {
            size_t bad = 0, first = h.size();
            for (size_t i = 0; i < h.size(); i++) if (h[i] != g_ref[i]) { bad++; if (first == h.size()) first = i; }
            printf("%-22s WRONG RESULT: %zu of %zu words differ, first at %zu (lane %zu): %08x vs %08x\\n", name, bad, h.size(), first, first / 2, h[first], g_ref[first]);
            return;
        }
    }
    printf("%-22s", name);
    for (int wps : {1, 2, 3, 4, 6})
    {
        // dynamic LDS pins exactly `wps` workgroups (of one wave per SIMD each) on every CU: 160 KiB per CU
        if (wps * 256 % threads) { continue; }
        if (threads == 768 && wps != 6 && wps != 3) { continue; }
        const int per_cu = wps * 256 / threads;                      // workgroups per CU
        const int lds = per_cu == 1 ? (96 << 10) : per_cu == 2 ? (64 << 10) : per_cu == 3 ? (48 << 10) : per_cu == 4 ? (36 << 10) : (24 << 10);
        const int blocks = 256 * per_cu;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, d, rc, perms); (void)hipDeviceSynchronize();
        float best = 1e30f;
        for (int r = 0; r < 5; r++)
        {
            (void)hipEventRecord(a); hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, d, rc, perms); (void)hipEventRecord(b);
            (void)hipEventSynchronize(b); float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        // chip-wide permutation rate, and issue cycles per VALU instruction at a nominal 2.4 GHz (4 560 per permutation)
        const double states = (double)blocks * threads * perms;
        const double cyc = best * 1e-3 * 2.4e9 / ((double)wps * perms * 4560.0);
        printf(" | %dw %6.3f ms %5.2f Gp/s %4.2f c/i", wps, best, states / (best * 1e-3) / 1e9, cyc);
    }
    printf("\\n");
}

int main()
{
    uint32_t* d; (void)hipMalloc(&d, 64 << 20);
    uint32_t* rc; (void)hipMalloc(&rc, sizeof kRC); (void)hipMemcpy(rc, kRC, sizeof kRC, hipMemcpyHostToDevice);
    run("compiler (unfolded)", k_compiler, d, rc, true);
    run("compiler (folded)", k_compiler_folded, d, rc, false);
''')
    for n in names:
        parts.append(f'    run("{n}", k_{n}, d, rc, false, {thr[n]});')
    parts.append('''    run("compiler (unfolded)", k_compiler, d, rc, false);
    (void)hipFree(d); (void)hipFree(rc);
    return 0;
}''')
    with open(os.path.join(here, "ubench7.hip"), "w") as f:
        f.write("\n".join(parts) + "\n")


if __name__ == "__main__":
    main()
