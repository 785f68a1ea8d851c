#!/usr/bin/env python3
"""Summarise the `tools/gpu_run.sh pmc` passes into profiles/pmc_traffic.json (read by bench.py for
roofline.traffic, stamped with the hash of the kernel sources it was collected for) and a per-round CSV.

  python tools/pmc_summary.py <workload> [round-tag]

Inputs: gpurun_out/pmc_<workload>/{FETCH_SIZE,WRITE_SIZE}_counter_collection.csv (rocprofv3 --pmc, one
counter per pass, unit KB).  The 1 GiB device copy at the start of tools/pmc_run.py calibrates the gfx950
corrections (MI355X_MICROARCH.md, HBM section): true bytes / counted bytes per counter.  Values are per
STEP: the sum over all launches of a kernel divided by the number of steps pmc_run.py ran (a kernel may
run once per prime)."""
import csv, json, os, re, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bench
workload = sys.argv[1] if len(sys.argv) > 1 else "c2"
tag = sys.argv[2] if len(sys.argv) > 2 else "r02"
batch = int(os.environ.get("SE_PMC_BATCH", bench.WORKLOADS[workload][3]))
STEPS = 3
src = os.path.join(ROOT, "gpurun_out", "pmc_" + workload)

def short(name):
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]

tot, cal = {}, {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = defaultdict(list)
    with open(os.path.join(src, f"{ctr}_counter_collection.csv")) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == ctr:
                acc[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    tot[ctr] = {k: sum(v) for k, v in acc.items()}
    cal[ctr] = max(acc["__amd_rocclr_copyBuffer"])   # the 1 GiB calibration copy (others are tiny)

GiB = float(1 << 30)
cal_f = GiB / (cal["FETCH_SIZE"] * 1024)
cal_w = GiB / (cal["WRITE_SIZE"] * 1024)
kernels = sorted(k for k in tot["FETCH_SIZE"] if k.startswith("seamd::"))
with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_counters_{workload}.csv"), "w") as f:
    f.write("kernel,FETCH_SIZE_KB_per_step,WRITE_SIZE_KB_per_step,calibration_fetch,calibration_write\n")
    for k in kernels:
        f.write(f"{k},{tot['FETCH_SIZE'][k] / STEPS:.1f},{tot['WRITE_SIZE'].get(k, 0.0) / STEPS:.1f},{cal_f:.4f},{cal_w:.4f}\n")

path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
out = json.load(open(path)) if os.path.exists(path) else {}
out["_method"] = (
    "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes with --kernel-trace "
    "(tools/gpu_run.sh pmc, tools/pmc_run.py, summarised by tools/pmc_summary.py); unit KB; calibrated on a "
    "1 GiB device copy in the same run (gfx950 counts 128-B fetch requests at 64 B, MI355X_MICROARCH.md HBM "
    "section): per-workload factors under _calibration. Bytes per STEP = sum over the kernel's launches / steps.")
out.setdefault("_calibration", {})[workload] = {"fetch_scale": cal_f, "write_scale": cal_w}
out.setdefault("_source_sha256", {})[workload] = bench.kernel_source_hash()
entry = {}
for k in kernels:
    m = re.match(r"seamd::(k_\w+)", k)
    if not m:
        continue
    fb = tot["FETCH_SIZE"][k] / STEPS * 1024 * cal_f
    wb = tot["WRITE_SIZE"].get(k, 0.0) / STEPS * 1024 * cal_w
    if fb + wb < 1e6:
        continue      # key-setup kernels
    entry[m.group(1)] = {"batch": batch, "hbm_bytes_per_step": int(round(fb + wb)),
                         "fetch_bytes": int(round(fb)), "write_bytes": int(round(wb))}
out[workload] = entry
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out[workload], indent=1))
