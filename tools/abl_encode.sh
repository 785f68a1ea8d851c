#!/bin/bash
# timing-only ablations of the transform kernels (results are WRONG with an ABL flag set)
mkdir -p gpurun_out
for w in c5 c2; do ( SE_BENCH_SKIP_STATUS=1 timeout 600 python bench.py --steps 5 --warmup 1 --workload $w --no-cpu-baseline --others none ) 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', d['config']['workload'][:2], '%.2f ms/step' % d['ms_per_step'], {k['kernel'][2:]: round(k['ms_per_step'], 3) for k in d['roofline']['kernels']})
"; done
