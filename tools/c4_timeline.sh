#!/bin/bash
# kernel timeline of one C4 step (rocprofv3 --kernel-trace): start / end of every launch relative to the step
cd "$(dirname "$0")/.."
export TMPDIR=/tmp; rm -rf /tmp/c4tl
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/c4tl -o tl -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --workload ${TL_WL:-c4} --no-cpu-baseline --others none ) > gpurun_out/c4_timeline.log 2>&1
python - <<'PY'
import csv, glob, os
f = glob.glob('/tmp/c4tl/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'seamd' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last full step: walk back from the end to the last k_sample_cbd
idx = max(i for i, r in enumerate(rows) if 'k_sample_cbd' in r['Kernel_Name'])
# the profiled (event-timed) steps come last; take the step before them: find cbd launches
cbds = [i for i, r in enumerate(rows) if 'k_sample_cbd' in r['Kernel_Name']]
i0 = (cbds[-5] if os.environ.get('TL_TWO') and len(cbds) >= 5 else cbds[-4]) if len(cbds) >= 4 else cbds[0]
i1 = cbds[-3] if len(cbds) >= 4 else len(rows)
t0 = min(int(r['Start_Timestamp']) for r in rows[i0 - 1:i1])
for r in rows[max(0, i0 - 1):i1]:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    print('%-34s %9.3f -> %9.3f ms  (%.3f)' % (r['Kernel_Name'].split('(')[0].replace('void seamd::', '')[:34], s / 1e6, e / 1e6, (e - s) / 1e6))
PY
