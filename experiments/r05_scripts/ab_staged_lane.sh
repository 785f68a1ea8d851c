#!/bin/bash
# Round 5, VERDICT r4 item 1: the staged-lane sampler phase (debug flag 2048: k_bulk_lane chains + the candidate window
# in k_candidates beside them + light resolves, the FUSED kernel after) against the one-launch lane chain
# (k_sample_uniform, flag 8192) at the BASELINE batch, alternating inside ONE gpurun call.
#   usage: bash tools/ab_staged_lane.sh [workload=c2] [reps=3]      (on the GPU box)
cd "$(dirname "$0")/.."
W=${1:-c2}; REPS=${2:-3}
run() {   # label, debug flags, extra env
  env $3 SE_BENCH_DEBUG_FLAGS=$2 python bench.py --steps ${AB_STEPS:-20} --warmup 5 --workload $W --no-cpu-baseline --others none ${AB_ARGS:-} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); r = d['roofline']
print('%-22s' % '$1', '$W', '%.3f ms' % d['ms_per_step'], '%.3f M/s' % (d['value'] / 1e6), 'clk', (r.get('sampled_clock') or {}).get('mean_mhz'), {k['kernel'][2:]: round(k['ms_per_step'], 3) for k in r['kernels']})"
}
for rep in $(seq $REPS); do
  run lane_chain 8192 ""
  run staged_lane_s3 2048 "SE_AMD_WINDOW_SIGMA=3"
  run paired_chains 32768 "SE_AMD_WINDOW_SIGMA=3"
  run paired_hog $((32768+65536)) "SE_AMD_WINDOW_SIGMA=3"
  run paired_prio $((32768+131072)) "SE_AMD_WINDOW_SIGMA=3"
  run paired_hog_prio $((32768+65536+131072)) "SE_AMD_WINDOW_SIGMA=3"
done
