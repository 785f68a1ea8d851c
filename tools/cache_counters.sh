#!/bin/bash
# L1 (TCP) / L2 (TCC) request counters of one workload's kernels: CC_WL=c5 bash tools/cache_counters.sh
cd "$(dirname "$0")/.."
W=${CC_WL:-c5}; export TMPDIR=/tmp; mkdir -p gpurun_out/cc_$W
i=0
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_EA_RDREQ_sum"; do
  i=$((i+1)); rm -rf /tmp/cc_$i
  ( cd /tmp && timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/cc_$i -o cc -- python $GRAFT_REPO_ROOT/tools/pmc_run.py $W ) > gpurun_out/cc_$W/run_$i.log 2>&1
  find /tmp/cc_$i -name "*counter_collection.csv" -exec cp {} gpurun_out/cc_$W/set${i}.csv \;
done
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in sorted(glob.glob('gpurun_out/cc_$W/set*.csv')):
    seen = set()
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0].replace('void ', '')
        if 'seamd' not in k and 'copyBuffer' not in k: continue
        tot[k][row['Counter_Name']] += float(row['Counter_Value'])
for k, d in tot.items():
    print(k)
    for c, v in sorted(d.items()): print('   %-40s %.4g per step' % (c, v / 3))
PY
