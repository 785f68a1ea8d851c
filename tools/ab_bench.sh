#!/bin/bash
# A/B several builds of the library in ONE gpurun call (box-to-box variance is larger than most kernel
# changes): AB_LIBS names files seal-embedded_amd/lib/<name>.so ("new" = the current product library),
# built with `make -C seal-embedded_amd/csrc VARIANT=<name> ABL="-D..."`; alternating, two rounds.
cd "$(dirname "$0")/.."
L=seal-embedded_amd/lib
cp $L/libseal_embedded_amd.so /tmp/new.so
for rep in 1 2; do
for which in ${AB_LIBS:-new old}; do
  if [ $which = new ]; then cp /tmp/new.so $L/libseal_embedded_amd.so; else cp $L/$which.so $L/libseal_embedded_amd.so || continue; fi
  for w in ${AB_WL:-c2 c3 c5}; do
    python bench.py --steps 8 --warmup 2 --workload $w --no-cpu-baseline --others none ${AB_ARGS:-} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('%-8s' % '$which', '$w', '%.3f ms' % d['ms_per_step'], {k['kernel'][2:]: round(k['ms_per_step'], 3) for k in d['roofline']['kernels']})"
  done
done
done
cp /tmp/new.so $L/libseal_embedded_amd.so
