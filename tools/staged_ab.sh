for rep in 1 2; do
for cfg in "c4 0 0" "c4 0 1024" "c2 32768 0" "c2 32768 1024" "c2 16384 0" "c2 16384 1024" "c2 8192 0" "c2 8192 1024"; do
set -- $cfg
SE_BENCH_DEBUG_FLAGS=$3 python bench.py --steps 8 --warmup 2 --workload $1 --batch $2 --no-cpu-baseline --others none 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('$1 B=$2 flags=$3', '%.3f ms' % d['ms_per_step'], {k['kernel'][2:]: round(k['ms_per_step'], 3) for k in d['roofline']['kernels']})"
done; done
