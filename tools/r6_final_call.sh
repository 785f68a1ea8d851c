export TMPDIR=/tmp
O=gpurun_out/r6final; mkdir -p $O
( timeout 2400 python -m pytest tests -m gpu -q --tb=short --timeout=900 ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
( timeout 900 python bench.py ) > $O/bench.log 2>&1; tail -c 700 $O/bench.log
for w in x1 x2; do ( timeout 600 python bench.py --workload $w --steps 10 --warmup 3 --others none --no-cpu-baseline ) > $O/bench_$w.log 2>&1; tail -1 $O/bench_$w.log | cut -c1-250; done
( timeout 900 bash tools/ref_bench.sh ) > $O/ref_bench.log 2>&1; grep -A1 "^== " $O/ref_bench.log | head -40
( timeout 600 bash tools/api_latency.sh ) > $O/api_latency.log 2>&1; tail -12 $O/api_latency.log
( timeout 300 python tools/lower_sym_latency.py ) > $O/lower_sym_latency.log 2>&1; tail -8 $O/lower_sym_latency.log
