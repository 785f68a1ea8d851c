#!/bin/bash
# SQ counters of one workload under a debug-flag setting, summarised per kernel and per step:
#   bash tools/r6_sq.sh <workload> <flags> <tag>     (-> gpurun_out/sq_<tag>.txt)
cd "$(dirname "$0")/.."
W=$1; FLAGS=$2; TAG=$3
mkdir -p gpurun_out
export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" \
           ${SQ_EXTRA:+"$SQ_EXTRA"}; do
  i=$((i+1)); rm -rf /tmp/sq_$i
  ( cd /tmp && SE_PMC_FLAGS=$FLAGS timeout 900 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/sq_$i -o sq -- python $GRAFT_REPO_ROOT/tools/pmc_run.py $W ) > gpurun_out/sq_${TAG}_$i.log 2>&1
  mkdir -p gpurun_out/sq_$TAG
  find /tmp/sq_$i -name "*counter_collection.csv" -exec cp {} gpurun_out/sq_$TAG/set${i}_counter_collection.csv \;
done
python - "$TAG" <<'PY' | tee gpurun_out/sq_$TAG.txt
import csv, glob, collections, sys
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in sorted(glob.glob("gpurun_out/sq_%s/*.csv" % tag)):
    for r in csv.DictReader(open(f)):
        nm = r['Kernel_Name']
        if 'seamd::' not in nm: continue
        key = nm.split('(')[0].replace('void ', '').replace('seamd::', '').strip()
        acc[key][r['Counter_Name']] += float(r['Counter_Value']) / 3
print("# SQ counters per STEP,", tag)
for k, d in acc.items():
    if d.get('SQ_INSTS_VALU', 0) < 1e6: continue
    print(k)
    for c, v in d.items(): print("   %-28s %16.0f" % (c, v))
PY
rm -rf gpurun_out/sq_$TAG
