#!/usr/bin/env python3
"""Summarise gpurun_out/sq_<workload>/*.csv (SQ counter passes of tools/gpu_run.sh sq): per kernel, every
counter summed over its launches and divided by the steps pmc_run.py ran -> per STEP.  Writes
profiles/<tag>_sq_counters_<workload>.txt and profiles/sq_counters.json (read by bench.py for the VALU
bound, stamped with the hash of the kernel sources).

  python tools/sq_summary.py <workload> [round-tag]"""
import csv, glob, collections, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bench
workload = sys.argv[1] if len(sys.argv) > 1 else "c2"
tag = sys.argv[2] if len(sys.argv) > 2 else "r02"
batch = int(os.environ.get("SE_PMC_BATCH", bench.WORKLOADS[workload][3]))
STEPS = 3
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "sq_" + workload, "*.csv"))):
    for r in csv.DictReader(open(f)):
        nm = r['Kernel_Name']
        if 'seamd::' not in nm or 'ntt_polys' in nm or 'make_pairs' in nm or 'reduce_small' in nm: continue
        key = nm.split('(')[0].replace('void ', '').replace('seamd::', '').strip().split('<')[0]
        acc[key][r['Counter_Name']] += float(r['Counter_Value']) / STEPS
lines = []
for k, d in acc.items():
    lines.append(k)
    for c, v in d.items():
        lines.append("   %-28s %16.0f" % (c, v))
txt = "\n".join(lines)
print(txt)
open(os.path.join(ROOT, "profiles", f"{tag}_sq_counters_{workload}.txt"), "w").write(
    f"# SQ counters per STEP ({workload}, batch {batch}); rocprofv3 --pmc passes, tools/gpu_run.sh sq\n" + txt + "\n")
path = os.path.join(ROOT, "profiles", "sq_counters.json")
out = json.load(open(path)) if os.path.exists(path) else {}
out.setdefault("_source_sha256", {})[workload] = bench.kernel_source_hash()
out[workload] = {k: {"batch": batch, "valu_wave_insts_per_step": int(d.get("SQ_INSTS_VALU", 0)),
                     "lds_bank_conflict_cycles_per_step": int(d.get("SQ_LDS_BANK_CONFLICT", 0)),
                     "lds_idx_active_cycles_per_step": int(d.get("SQ_LDS_IDX_ACTIVE", 0)),
                     "wave_cycles_per_step": int(d.get("SQ_WAVE_CYCLES", 0)),
                     "wait_inst_any_per_step": int(d.get("SQ_WAIT_INST_ANY", 0)),
                     "active_inst_valu_per_step": int(d.get("SQ_ACTIVE_INST_VALU", 0))}
                 for k, d in acc.items() if d.get("SQ_INSTS_VALU")}
json.dump(out, open(path, "w"), indent=1)
