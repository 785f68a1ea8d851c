#!/usr/bin/env python3
"""valu_mix.py -- opcode-weighted VALU issue cost of the product kernels.

`bench.py`'s first VALU bound charges every wave instruction 4 SIMD cycles.  That is an estimate: on gfx950
`v_xor_b32 / v_add_u32 / v_sub_u32` (and whatever else tools/ubench2 shows above the 4-cycle class) issue
faster, `v_bitop3_b32` a little faster.  This tool

  1. reads the per-opcode issue rates measured by tools/ubench2 on the MI355X (a log of its output; rates are
     normalised to v_mul_lo_u32 = 4.0 cycles per wave instruction, so the clock the log was taken at drops out),
  2. compiles the kernel sources to ISA (hipcc -S, gfx950) and builds, per kernel, the instruction mix of its
     hot code: basic blocks weighted by trip counts of the loops around them (backward branches; the trip count
     per nesting level is a per-kernel constant below -- the Keccak round loops run 11-12 times, the prime loop
     of the fused kernels nprimes times),
  3. writes profiles/valu_mix.json: for every kernel the share of each issue class and the mean issue cycles per
     VALU wave instruction ("cpi").  bench.py multiplies the DYNAMIC instruction count of each kernel
     (SQ_INSTS_VALU, profiles/sq_counters.json) with its cpi:  floor_ms_weighted = sum_k insts_k * cpi_k / SIMDs / clock.

  python tools/valu_mix.py gpurun_out/ubench2_r04.log            (writes profiles/valu_mix.json)
"""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# ubench2 kernel name -> the mnemonics it stands for
UBENCH_OPS = {
    "k_xor": ["v_xor_b32"], "k_add": ["v_add_u32"], "k_sub": ["v_sub_u32"], "k_subrev": ["v_subrev_u32"],
    "k_and": ["v_and_b32"], "k_or": ["v_or_b32"], "k_not": ["v_not_b32"], "k_mov": ["v_mov_b32"],
    "k_bitop3": ["v_bitop3_b32"], "k_alignbit": ["v_alignbit_b32"], "k_min": ["v_min_u32"], "k_max": ["v_max_u32"],
    "k_lshl": ["v_lshlrev_b32"], "k_lshr": ["v_lshrrev_b32"], "k_ashr": ["v_ashrrev_i32"],
    "k_mul_lo": ["v_mul_lo_u32"], "k_mul_hi": ["v_mul_hi_u32"], "k_add3": ["v_add3_u32"], "k_or3": ["v_or3_b32"],
    "k_and_or": ["v_and_or_b32"], "k_lshl_or": ["v_lshl_or_b32"], "k_lshl_add": ["v_lshl_add_u32"],
    "k_bfe": ["v_bfe_u32"], "k_bcnt": ["v_bcnt_u32_b32"], "k_perm": ["v_perm_b32"], "k_bfi": ["v_bfi_b32"],
    "k_add_co": ["v_add_co_u32", "v_sub_co_u32", "v_subrev_co_u32"], "k_addc_co": ["v_addc_co_u32", "v_subb_co_u32"],
    "k_fma64": ["v_fma_f64"], "k_mul64": ["v_mul_f64"], "k_add64": ["v_add_f64"], "k_max64": ["v_max_f64"],
    "k_rndne64": ["v_rndne_f64", "v_trunc_f64", "v_floor_f64", "v_ceil_f64"],
    "k_mad6432": ["v_mad_u64_u32"], "k_cvt_i32_f64": ["v_cvt_i32_f64"], "k_cvt_f64_i32": ["v_cvt_f64_i32"],
    "k_lshladd64": ["v_lshl_add_u64"], "k_lshl64": ["v_lshlrev_b64"], "k_mov_dpp": ["v_mov_b32_dpp"],
    "k_add_f32": ["v_add_f32"], "k_mul_f32": ["v_mul_f32"], "k_fma32": ["v_fma_f32"], "k_cndmask": ["v_cndmask_b32"],
}
# loop trip count per nesting level, by kernel-name prefix (first match); default: the Keccak round loops
TRIPS = [("k_encode_encrypt", 3.0), ("k_encode_rns", 3.0), ("k_ntt_fuse", 1.0), ("k_ntt_polys", 1.0)]
DEFAULT_TRIP = 12.0
# source file -> (key bench.py looks up: "<kernel>@<workload mode or logn>", mangled-name fragment)
KERNELS = {
    "encode_encrypt": [("k_encode_encrypt@sym12", "k_encode_encryptILi12ELi0E"),
                       ("k_encode_encrypt@asym12", "k_encode_encryptILi12ELi1E"),
                       ("k_encode_encrypt@encode12", "k_encode_encryptILi12ELi2E"),
                       ("k_encode_encrypt@sym10", "k_encode_encryptILi10ELi0E"),
                       ("k_encode_rns@sym14", "k_encode_rnsILi14ELb1E"), ("k_ntt_fuse@sym14", "k_ntt_fuseILi14ELi0E"),
                       ("k_encode_rns@sym13", "k_encode_rnsILi13ELb1E"), ("k_ntt_fuse@sym13", "k_ntt_fuseILi13ELi0E"),
                       ("k_encode_encrypt@asym14", "k_encode_encryptILi14ELi1E")],
    "samplers": [("k_sample_uniform@12", "k_sample_uniformILi12ELi512ELb0E"),
                 ("k_sample_uniform@14", "k_sample_uniformILi14ELi512ELb0E"),
                 ("k_sample_uniform@13", "k_sample_uniformILi13ELi512ELb0E"),
                 ("k_sample_cbd", "k_sample_cbd"), ("k_sample_ternary", "k_sample_ternaryILi512E"),
                 ("k_sample_ternary_window", "k_sample_ternary_window"),
                 ("k_bulk_pair@14", "k_bulk_pairILi14E"), ("k_bulk_pair@13", "k_bulk_pairILi13E"),
                 ("k_candidates", "k_candidates"),
                 ("k_resolve_light@14", "k_resolve_lightILi14E"), ("k_resolve_light@13", "k_resolve_lightILi13E")],
}


def issue_cycles(log_path):
    rate = {}
    for ln in open(log_path):
        m = re.match(r"^(k_\w+)\s+[\d.]+ ms\s+([\d.]+) lane-ops", ln)
        if m:
            rate[m.group(1)] = float(m.group(2))
    ref = rate["k_mul_lo"]
    cyc = {}
    if "k_cmp_gt" in rate:
        c_cmp = 4.0 * ref / rate["k_cmp_gt"]
        for op in ("v_cmp_gt_u32", "v_cmp_lt_u32", "v_cmp_ge_u32", "v_cmp_le_u32", "v_cmp_eq_u32", "v_cmp_ne_u32",
                   "v_cmp_gt_i32", "v_cmp_lt_i32", "v_cmp_eq_u64", "v_cmp_ne_u64", "v_cmp_gt_u64", "v_cmp_lt_u64"):
            cyc[op] = round(c_cmp, 3)
        if "k_cmp_cnd" in rate:      # the pair kernel issues one compare + one select per counted op
            cyc["v_cndmask_b32"] = round(max(4.0 * ref / rate["k_cmp_cnd"] - c_cmp, 1.0), 3)
    for k, ops in UBENCH_OPS.items():
        if k in rate and rate[k] > 0 and k != "k_cndmask":      # ubench2's cndmask chain serialises on vcc: not a rate
            for op in ops:
                cyc[op] = round(4.0 * ref / rate[k], 3)
    # mixed streams of ubench2: measured cycles per instruction against what the single-opcode table predicts for
    # that mix.  k_mixind / k_runs* (neighbouring instructions independent) reach the prediction; k_mix_keccak /
    # k_mix_ntt (every instruction consumes its predecessor's result) do not: ~1.2 cycles per dependent pair.
    mixes = {}
    keccak_ops, ntt_ops = ("v_xor_b32", "v_bitop3_b32", "v_alignbit_b32"), ("v_sub_u32", "v_min_u32", "v_mul_hi_u32", "v_mul_lo_u32", "v_add3_u32")
    for key, ops in (("k_mix_keccak", keccak_ops), ("k_mixind", keccak_ops), ("k_runs1", keccak_ops),
                     ("k_runs2", keccak_ops), ("k_runs4", keccak_ops), ("k_runs8_keccak", keccak_ops),
                     ("k_mix_ntt", ntt_ops)):
        if key in rate and all(o in cyc for o in ops):
            measured = 4.0 * ref / rate[key]
            predicted = sum(cyc[o] for o in ops) / len(ops)
            mixes[key] = {"measured_cycles_per_inst": round(measured, 3), "predicted": round(predicted, 3),
                          "ratio": round(measured / predicted, 4)}
    rate["_mixes"] = mixes
    return cyc, rate


def compile_asm(name):
    csrc = os.path.join(ROOT, "seal-embedded_amd", "csrc")
    out = f"/tmp/valu_mix_{name}.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950",
                           "-I" + csrc, "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
                           os.path.join(csrc, "kernels", name + ".hip"), "-o", out], stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def kernel_mix(lines, frag, cyc):
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(frag) + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    name = re.match(r"^(_Z\w+):", lines[start]).group(1)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    short = re.sub(r"\(.*", "", dem).replace("void seamd::", "")
    body = lines[start + 1:end + 1]
    # Basic blocks and the loops around them, from the compiler's own block comments
    # ("=>This Inner Loop Header: Depth=2", "in Loop: Header=BB30_5 Depth=2", "Parent Loop BB30_4 Depth=1").
    # Trip counts: the fused / split transform kernels have ONE loop, the prime loop (TRIPS); in the sampler
    # kernels an innermost loop that holds >= 150 VALU instructions is a Keccak round loop (2 rounds per
    # iteration: 12 trips), a loop AROUND such a loop is a squeeze / candidate loop (121 steps at n = 4096; only
    # its ratio to the inner loop matters), anything else is a short data-dependent loop (reject lists, redraw
    # rounds: ~3 trips)
    insts, inst_loops = [], []
    cur_loops, pending_label = (), None
    # loops INSIDE an inline-asm block (keccak_sync.cuh: "s_movk_i32 s30, <trips>" / "1:" ... "s_cbranch_scc1 1b") carry
    # no compiler annotation: they get a pseudo header with the trip count the block states itself
    asm_loop, asm_trips, last_movk, n_asm = None, {}, None, 0
    for l in body:
        t = l.strip()
        ma = re.match(r"^s_movk_i32 s30, (\w+)", t)
        if ma:
            last_movk = float(int(ma.group(1), 0))
        if re.match(r"^\d+:$", t):
            n_asm += 1
            asm_loop = "asm%d" % n_asm
            asm_trips[asm_loop] = last_movk if last_movk else DEFAULT_TRIP
            continue
        m = re.match(r"^\.(LBB\w+):", t)
        if m:
            pending_label, cur_loops = m.group(1)[1:], ()
            if ";" in t:
                t = t[t.index(";"):]
            else:
                continue
        if t.startswith(";"):
            if pending_label is not None:
                mh = re.search(r"Loop Header: Depth=(\d+)", t)
                mi = re.search(r"in Loop: Header=(BB\w+) Depth=(\d+)", t)
                mp = re.search(r"Parent Loop (BB\w+) Depth=(\d+)", t)
                if mh:
                    cur_loops += (pending_label[1:] if pending_label.startswith("L") else pending_label,)
                elif mi:
                    cur_loops += (mi.group(1),)
                elif mp:
                    cur_loops += (mp.group(1),)
            continue
        if not t or t.startswith((".", "//")):
            continue
        pending_label = None
        insts.append(t)
        inst_loops.append(cur_loops + ((asm_loop,) if asm_loop else ()))
        if asm_loop and re.match(r"^s_cbranch_scc1 \d+b", t):
            asm_loop = None
    is_valu = [t.split()[0].startswith("v_") for t in insts]
    size = collections.Counter()
    for v, ls in zip(is_valu, inst_loops):
        if v:
            for h in ls:
                size[h] += 1
    nests = set()          # (outer, inner) pairs
    for ls in inst_loops:
        for a in ls:
            for b2 in ls:
                if a != b2:
                    nests.add((a, b2))
    children = lambda h: [b2 for (a, b2) in nests if a == h and size[b2] < size[h]]
    fixed = next((v for p, v in TRIPS if short.startswith(p)), None)
    keccak = {h for h in size if size[h] >= 150 and not any(size[c] >= 150 for c in children(h))}
    trip_of = {}
    for h in size:
        if h in asm_trips:
            trip_of[h] = asm_trips[h]
        elif fixed is not None:
            trip_of[h] = fixed
        elif h in keccak:
            trip_of[h] = 12.0
        elif any(c in keccak for c in children(h)):
            trip_of[h] = 121.0
        else:
            trip_of[h] = 3.0
    weight = []
    for ls in inst_loops:
        w = 1.0
        for h in ls:
            w *= trip_of[h]
        weight.append(w)
    trip = fixed if fixed is not None else DEFAULT_TRIP
    hist = collections.Counter()
    for i, t in enumerate(insts):
        op = t.split()[0]
        if not op.startswith("v_"):
            continue
        op = re.sub(r"_(e32|e64|sdwa)$", "", op)
        if op.endswith("_dpp"):
            op = "v_mov_b32_dpp" if op.startswith("v_mov") else op[:-4]
        hist[op] += weight[i]
    total = sum(hist.values())
    cls = collections.Counter()
    cycles = 0.0
    unknown = collections.Counter()
    for op, c in hist.items():
        cy = cyc.get(op)
        if cy is None:
            unknown[op] += c
            cy = 4.0
        cycles += c * cy
        cls["fast (< 3 cycles)" if cy < 3.0 else "bitop3-class (3 - 3.8)" if cy < 3.8 else "full (>= 3.8)"] += c
    return short, {
        "weighted_valu_insts_per_wave": round(total, 1),
        "cycles_per_inst": round(cycles / total, 4),
        "share": {k: round(v / total, 4) for k, v in sorted(cls.items())},
        "top": {op: round(c / total, 4) for op, c in hist.most_common(10)},
        "unmeasured_opcodes_share": round(sum(unknown.values()) / total, 4),
        "loop_trip_per_level": trip,
    }


def main():
    log = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "ubench2_r04.log")
    cyc, rate = issue_cycles(log)
    import bench
    out = {"_source_sha256": bench.kernel_source_hash(), "_rates_from": os.path.basename(log),
           "_note": "issue cycles per wave64 VALU instruction on one SIMD, tools/ubench2 (8 waves per SIMD, 8 independent "
                    "chains per lane), normalised to v_mul_lo_u32 = 4.0; opcodes ubench2 does not cover count 4.0",
           "issue_cycles": dict(sorted(cyc.items())), "mixed_streams": rate.get("_mixes", {}), "kernels": {}}
    for src, frags in KERNELS.items():
        lines = compile_asm(src)
        for key, f in frags:
            try:
                name, mix = kernel_mix(lines, f, cyc)
            except StopIteration:
                print("not found:", f)
                continue
            mix["kernel"] = name
            out["kernels"][key] = mix
            print(f"{key:28s} {name:40s} cpi {mix['cycles_per_inst']:.3f}  {mix['share']}  unmeasured {mix['unmeasured_opcodes_share']}")
    with open(os.path.join(ROOT, "profiles", "valu_mix.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
