export TMPDIR=/tmp
mkdir -p gpurun_out/r5
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --tb=short --timeout=600 -k "staged_lane or staged_sampler" ) > gpurun_out/r5/pytest_staged_lane.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5/pytest_staged_lane.log; tail -5 gpurun_out/r5/pytest_staged_lane.log
( timeout 900 bash tools/ab_staged_lane.sh c2 3 ) > gpurun_out/r5/ab_staged_lane_c2.log 2>&1; cat gpurun_out/r5/ab_staged_lane_c2.log
cd /tmp && rm -rf /tmp/prof && ( SE_BENCH_DEBUG_FLAGS=${PROF_FLAGS:-2048} timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o sl -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --workload c2 --no-cpu-baseline --others none ) > $GRAFT_REPO_ROOT/gpurun_out/r5/prof_sl.log 2>&1
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5/prof_sl; find /tmp/prof -name "*kernel_stats*.csv" -exec cp {} gpurun_out/r5/prof_sl/ \; ; find /tmp/prof -name "*kernel_trace*.csv" -size -20000k -exec cp {} gpurun_out/r5/prof_sl/ \; ; head -14 gpurun_out/r5/prof_sl/*kernel_stats*.csv
rocm-smi --showcomputepartition 2>&1 | tail -5
