for rep in 1 2; do for ch in 1 2 4 8 12; do SE_AMD_ASYM_CHUNKS=$ch python bench.py --steps 8 --warmup 2 --workload c3 --no-cpu-baseline --others none 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('chunks $ch', '%.3f ms' % d['ms_per_step'], {k['kernel'][2:]: round(k['ms_per_step'], 3) for k in d['roofline']['kernels']})"; done; done
