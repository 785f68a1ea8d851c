export TMPDIR=/tmp
mkdir -p gpurun_out/r5
( timeout 1500 python -m pytest tests/test_gpu_lower.py tests/test_gpu_reftests.py -m gpu -x -q --tb=short --timeout=900 ) > gpurun_out/r5/pytest_lower.log 2>&1; echo "lower pytest rc=$?"; tail -15 gpurun_out/r5/pytest_lower.log
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --tb=short --timeout=800 -k "ragged_shapes or staged" ) > gpurun_out/r5/pytest_watchdog.log 2>&1; echo "watchdog pytest rc=$?"; tail -8 gpurun_out/r5/pytest_watchdog.log
bash tools/ref_bench.sh > gpurun_out/r5/ref_bench.log 2>&1; head -30 gpurun_out/r5/ref_bench.log
SE_AMD_LOWER_SPECULATION=0 bash tools/ref_bench.sh > gpurun_out/r5/ref_bench_nospec.log 2>&1; head -8 gpurun_out/r5/ref_bench_nospec.log
AB_LIBS="new maxilp trackers bias0" AB_WL="c2 c3 c5" bash tools/ab_bench.sh > gpurun_out/r5/ab_sched_flags.log 2>&1; cat gpurun_out/r5/ab_sched_flags.log
