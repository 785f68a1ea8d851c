#!/bin/bash
# latency A/B of two library builds in one call: the relinked reference benches (sym, asym, uniform, ternary), the
# single-call API latencies and C1; AB_LIBS names files seal-embedded_amd/lib/<name>.so ("new" = the product library)
cd "$(dirname "$0")/.."
L=seal-embedded_amd/lib
cp $L/libseal_embedded_amd.so /tmp/new.so
for rep in 1 2; do
for which in ${AB_LIBS:-new prev}; do
  if [ $which = new ]; then cp /tmp/new.so $L/libseal_embedded_amd.so; else cp $L/$which.so $L/libseal_embedded_amd.so; fi
  echo "== $which"
  bash tools/ref_bench.sh 2>/dev/null | grep -E "^== |avg" | paste - - | grep -E "sym|uniform|ternary" | sed 's/^/   /'
  bash tools/api_latency.sh 2>/dev/null | grep latency_ms | sed 's/^/   /'
  python bench.py --workload c1 --steps 200 --warmup 20 --others none --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('   c1 %.4f ms' % d['ms_per_step'])"
done
done
cp /tmp/new.so $L/libseal_embedded_amd.so
