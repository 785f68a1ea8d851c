#!/bin/bash
# Round-6 evidence, ONE gpurun call (same box, same clock state for the line and the profile beside it):
#   1. pytest -m gpu (whole suite)     2. the default bench line      3. the DRIVER's command (--steps 20 --warmup 5,
#   C2 only) under rocprofv3 --kernel-trace --stats: its own JSON line (HIP events) and the rocprof averages come from the
#   SAME process     4. PMC traffic + SQ counter passes for C2..C5     5. soaks
export TMPDIR=/tmp
O=gpurun_out/r6ev; mkdir -p $O
( timeout 2400 python -m pytest tests -m gpu -q --tb=short --timeout=900 ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
( timeout 900 python bench.py ) > $O/bench.log 2>&1; echo "bench rc=$?" >> $O/bench.log; tail -2 $O/bench.log | cut -c1-300
cd /tmp && rm -rf /tmp/prof && ( timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r06_c2 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --others none ) > $GRAFT_REPO_ROOT/$O/prof_c2.log 2>&1
cd $GRAFT_REPO_ROOT; mkdir -p $O/prof; find /tmp/prof -name "*kernel_stats*.csv" -exec cp {} $O/prof/ \; ; head -6 $O/prof/*kernel_stats*.csv | cut -c1-200
cd /tmp && rm -rf /tmp/prof && ( timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r06 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline ) > $GRAFT_REPO_ROOT/$O/prof_all.log 2>&1
cd $GRAFT_REPO_ROOT; find /tmp/prof -name "*kernel_stats*.csv" -exec cp {} $O/prof/ \;
for W in c2 c3 c4 c5; do PMC_WL=$W bash tools/gpu_run.sh pmc > $O/pmc_$W.log 2>&1; PMC_WL=$W bash tools/gpu_run.sh sq > $O/sq_$W.log 2>&1; done
ls gpurun_out/pmc_c2 gpurun_out/sq_c2
( timeout 800 env FUZZ_SECONDS=600 python tools/fuzz_parity.py ) > $O/fuzz_parity.log 2>&1; tail -2 $O/fuzz_parity.log
( timeout 500 env FUZZ_SECONDS=400 python tools/fuzz_mid.py ) > $O/fuzz_mid.log 2>&1; tail -2 $O/fuzz_mid.log
