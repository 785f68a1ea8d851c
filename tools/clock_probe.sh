#!/bin/bash
# Sustained engine clock / power while a workload loops: CLOCK_WL="c2 c5" bash tools/clock_probe.sh
# (run on the GPU box: gpurun -- 'bash tools/clock_probe.sh > gpurun_out/clock.log 2>&1')
cd "$(dirname "$0")/.."
for w in ${CLOCK_WL:-c2 c3 c5}; do
  steps=${CLOCK_STEPS:-$([ $w = c5 ] && echo 600 || echo 2500)}
  echo "== $w"
  python bench.py --steps $steps --warmup 2 --workload $w --no-cpu-baseline --others none > /tmp/clk_$w.json 2>/dev/null &
  pid=$!
  sleep ${CLOCK_DELAY:-14}
  for i in 1 2 3 4 5 6; do
    kill -0 $pid 2>/dev/null || break
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|fclk|mclk" | tr -s ' ' | tr '\n' ';'
    echo
    sleep 0.5
  done
  wait $pid
  python -c "
import json,sys
d=json.loads(open('/tmp/clk_$w.json').readlines()[-1]); print('$w', 'ms_per_step', round(d['ms_per_step'],3))"
done
