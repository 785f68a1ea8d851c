#!/usr/bin/env python3
"""isa_histogram.py -- static instruction mix of one kernel, per basic block.

  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -S --cuda-device-only \
        -Iseal-embedded_amd/csrc -Iinclude seal-embedded_amd/csrc/kernels/encode_encrypt.hip -o /tmp/ee.s
  python tools/isa_histogram.py /tmp/ee.s k_encode_encryptILi12ELi0E [--blocks]

Prints the VALU / SALU / LDS / VMEM / other counts of every basic block (label to label) and the mnemonic
histogram of the whole kernel, heaviest first.  Loop bodies have to be weighted by hand (the prime loop of
the fused kernel runs nprimes times); the dynamic total to compare with is SQ_INSTS_VALU per wave
(profiles/sq_counters.json).
"""
import collections
import re
import sys


def classify(m):
    if m.startswith(("v_cmp", "v_")):
        return "valu"
    if m.startswith("ds_"):
        return "lds"
    if m.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if m.startswith("s_waitcnt") or m.startswith("s_nop") or m.startswith("s_barrier"):
        return "wait"
    if m.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    show_blocks = "--blocks" in sys.argv
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(pat) + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    blocks, cur = [], ["entry", collections.Counter(), collections.Counter()]
    total = collections.Counter()
    kinds = collections.Counter()
    for l in lines[start + 1:end + 1]:
        t = l.strip()
        if not t or t.startswith((";", ".", "//")):
            mlabel = re.match(r"^(\.LBB\w+):", t)
            if mlabel:
                blocks.append(cur)
                cur = [mlabel.group(1), collections.Counter(), collections.Counter()]
            continue
        m = t.split()[0]
        k = classify(m)
        cur[1][k] += 1
        cur[2][m] += 1
        total[m] += 1
        kinds[k] += 1
    blocks.append(cur)
    print("kernel lines", start, end, dict(kinds))
    if show_blocks:
        for name, k, h in blocks:
            if sum(k.values()) >= 20:
                top = ", ".join(f"{m}:{c}" for m, c in h.most_common(8))
                print(f"  {name:12s} {dict(k)}  {top}")
    for m, c in total.most_common(45):
        print(f"  {m:28s} {c}")


if __name__ == "__main__":
    main()
