#!/bin/bash
# Like tools/ab_bench.sh with longer timed regions (AB_STEPS, default 40 after 10 warm-up steps) and AB_REPS rounds:
# for changes of a few per cent on the box-noise level of the short form.
cd "$(dirname "$0")/.."
L=seal-embedded_amd/lib
cp $L/libseal_embedded_amd.so /tmp/new.so
for rep in $(seq 1 ${AB_REPS:-3}); do
for which in ${AB_LIBS:-new old}; do
  if [ $which = new ]; then cp /tmp/new.so $L/libseal_embedded_amd.so; else cp $L/$which.so $L/libseal_embedded_amd.so || continue; fi
  for w in ${AB_WL:-c2}; do
    python bench.py --steps ${AB_STEPS:-40} --warmup 10 --workload $w --no-cpu-baseline --others none ${AB_ARGS:-} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('%-8s' % '$which', '$w', '%.3f ms' % d['ms_per_step'], {k['kernel'][2:]: round(k['ms_per_step'], 3) for k in d['roofline']['kernels']}, d['roofline']['sampled_clock']['mean_mhz'] if d['roofline'].get('sampled_clock') else None)"
  done
done
done
cp /tmp/new.so $L/libseal_embedded_amd.so
