#!/bin/bash
# A/B two builds of the library in ONE gpurun call (box-to-box variance is larger than most kernel
# changes): seal-embedded_amd/lib/old.so vs the current libseal_embedded_amd.so, alternating.
cd "$(dirname "$0")/.."
L=seal-embedded_amd/lib
cp $L/libseal_embedded_amd.so /tmp/new.so
for rep in 1 2; do
for which in new old; do
  cp $L/$which.so $L/libseal_embedded_amd.so 2>/dev/null || cp /tmp/new.so $L/libseal_embedded_amd.so
  [ $which = new ] && cp /tmp/new.so $L/libseal_embedded_amd.so
  for w in ${AB_WL:-c2 c3 c5}; do
    python bench.py --steps 8 --warmup 2 --workload $w --no-cpu-baseline ${AB_ARGS:-} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('$which', '$w', round(d['ms_per_step'], 3), 'ms', {k: round(v, 2) for k, v in d['roofline']['stage_ms_per_step'].items()})"
  done
done
done
cp /tmp/new.so $L/libseal_embedded_amd.so
