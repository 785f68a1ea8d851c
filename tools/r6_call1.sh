#!/bin/bash
# round 6, call 1: parity of the 8-points-per-thread fused kernel, then its A/B against the 16-point form at C5 and C2
# (debug flag 1 << 18 = 8 points, 1 << 19 = 16 points), for the 8-wave (64 VGPRs) and the 6-wave (80) builds.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=seal-embedded_amd/lib
LOG=gpurun_out/r6_call1.log
: > $LOG
[ -n "$SKIP_CHECK" ] || ( timeout 600 python tools/t8_check.py ) >> $LOG 2>&1
echo "t8_check rc=$?" >> $LOG
cp $L/libseal_embedded_amd.so /tmp/new.so
run() {  # lib flag workload
  python bench.py --steps ${AB_STEPS:-12} --warmup 3 --workload $3 --no-cpu-baseline --others none 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('%-6s %-8s' % ('$1', '$2'), '$3', '%.3f ms' % d['ms_per_step'], {k['kernel'][2:]: round(k['ms_per_step'], 3) for k in d['roofline']['kernels']}, 'clock', (d['roofline'].get('sampled_clock') or {}).get('mean_mhz'))"
}
for rep in $(seq 1 ${AB_REPS:-2}); do
  for lib in ${AB_LIBS:-new t8w6}; do
    if [ $lib = new ]; then cp /tmp/new.so $L/libseal_embedded_amd.so; else cp $L/$lib.so $L/libseal_embedded_amd.so; fi
    for w in c5 c2; do
      SE_BENCH_DEBUG_FLAGS=524288 run $lib pts16 $w >> $LOG 2>&1
      SE_BENCH_DEBUG_FLAGS=262144 run $lib pts8 $w >> $LOG 2>&1
    done
  done
done
cp /tmp/new.so $L/libseal_embedded_amd.so
if [ -n "$T8_EXHAUSTIVE" ]; then
  ( SE_AMD_TRANSFORM8=1 timeout 1500 python -m pytest tests/test_gpu_parity.py -q -k "test_full_size_properties_config2 and dispatch or test_full_size_encode_only_config5 or test_declined_plaintexts or test_nonfinite_values_through_every_path or test_encode_only_config5 or test_magnitude_classes" --timeout=1200 ) >> $LOG 2>&1
fi
cat $LOG
