export TMPDIR=/tmp
mkdir -p gpurun_out/r5
( timeout 1500 python -m pytest tests/test_gpu_lower.py tests/test_gpu_reftests.py -m gpu -x -q --tb=short --timeout=900 ) > gpurun_out/r5/pytest_lower.log 2>&1; echo "lower pytest rc=$?"; tail -3 gpurun_out/r5/pytest_lower.log
bash tools/ref_bench.sh > gpurun_out/r5/ref_bench.log 2>&1; head -12 gpurun_out/r5/ref_bench.log
python tools/lower_sym_latency.py 4096 3 6; python tools/lower_sym_latency.py 16384 6 4; SE_AMD_LOWER_SPECULATION=0 python tools/lower_sym_latency.py 16384 6 3
