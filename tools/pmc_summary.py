#!/usr/bin/env python3
"""Summarise the `tools/gpu_run.sh pmc` passes into profiles/pmc_traffic.json and
profiles/r01_pmc_counters.csv.

Inputs: gpurun_out/pmc/{FETCH_SIZE,WRITE_SIZE}_counter_collection.csv (rocprofv3 --pmc, one counter
per pass, unit KB).  The 1 GiB device copy at the start of tools/pmc_run.py calibrates the gfx950
corrections (MI355X_MICROARCH.md, HBM section): true bytes / counted bytes per counter."""
import csv, json, os, re, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "pmc")
workload, batch = (sys.argv[1] if len(sys.argv) > 1 else "c2"), 65536

def short(name):
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]

avg, cal = {}, {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = defaultdict(list)
    with open(os.path.join(src, f"{ctr}_counter_collection.csv")) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == ctr:
                acc[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    avg[ctr] = {k: sum(v) / len(v) for k, v in acc.items()}
    cal[ctr] = max(acc["__amd_rocclr_copyBuffer"])   # the 1 GiB calibration copy (others are tiny)

GiB = float(1 << 30)
cal_f = GiB / (cal["FETCH_SIZE"] * 1024)
cal_w = GiB / (cal["WRITE_SIZE"] * 1024)
kernels = sorted(k for k in avg["FETCH_SIZE"] if k.startswith("seamd::") or k.startswith("__amd_rocclr"))
with open(os.path.join(ROOT, "profiles", "r01_pmc_counters.csv"), "w") as f:
    f.write("kernel,FETCH_SIZE_KB_avg,WRITE_SIZE_KB_avg\n")
    for k in kernels:
        f.write(f"{k},{avg['FETCH_SIZE'][k]:.1f},{avg['WRITE_SIZE'].get(k, 0.0):.1f}\n")

path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
out = json.load(open(path)) if os.path.exists(path) else {}
out["_method"] = (
    "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes with --kernel-trace "
    "(tools/gpu_run.sh pmc, tools/pmc_run.py, summarised by tools/pmc_summary.py); unit KB; calibrated on a "
    f"1 GiB device copy in the same run: FETCH_SIZE read {1/cal_f:.3f} of the true bytes (gfx950 counts "
    f"128-B requests at 64 B, MI355X_MICROARCH.md HBM section) -> scaled by {cal_f:.3f}; WRITE_SIZE read "
    f"{1/cal_w:.3f} -> scaled by {cal_w:.3f}. Averages over 3 launches.")
entry = {}
for k in kernels:
    m = re.match(r"seamd::(k_\w+)", k)
    if not m:
        continue
    fb = avg["FETCH_SIZE"][k] * 1024 * cal_f
    wb = avg["WRITE_SIZE"].get(k, 0.0) * 1024 * cal_w
    if fb + wb < 1e6:
        continue      # key-setup kernels
    entry[m.group(1)] = {"batch": batch, "hbm_bytes_per_launch": int(round(fb + wb)),
                         "fetch_bytes": int(round(fb)), "write_bytes": int(round(wb))}
out[workload] = entry
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out[workload], indent=1))
