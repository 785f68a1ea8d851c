mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
( timeout 120 ./tools/ubench ) > gpurun_out/ubench.log 2>&1
( timeout 1500 python -m pytest tests -m gpu -q -x --timeout=600 ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
( timeout 600 python bench.py --steps 5 --warmup 2 ) > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
tail -5 gpurun_out/smoke.log; cat gpurun_out/ubench.log; tail -30 gpurun_out/pytest_gpu.log; tail -5 gpurun_out/bench.log
