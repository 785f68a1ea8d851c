export TMPDIR=/tmp
mkdir -p gpurun_out/r5
L=seal-embedded_amd/lib
cp $L/libseal_embedded_amd.so /tmp/new.so
cp $L/nttpair.so $L/libseal_embedded_amd.so
( SE_AMD_NTT_PAIR=1 timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --tb=line --timeout=900 -k "encrypt_sym_vs_oracle or all_pipeline_shapes or full_size_properties_config4 or staged_sampler_pipeline" ) > gpurun_out/r5/pytest_nttpair.log 2>&1; echo "nttpair pytest rc=$?"; tail -2 gpurun_out/r5/pytest_nttpair.log
run() { env $2 python bench.py --steps 10 --warmup 3 --workload $3 --no-cpu-baseline --others none 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('%-10s' % '$1', '$3', '%.3f ms' % d['ms_per_step'], {k['kernel'][2:]: round(k['ms_per_step'], 3) for k in d['roofline']['kernels']})"; }
for rep in 1 2 3; do for w in c4 x2; do run one_prime "" $w; run two_primes "SE_AMD_NTT_PAIR=1" $w; done; done > gpurun_out/r5/ab_nttpair.log 2>&1; cat gpurun_out/r5/ab_nttpair.log
rm -rf gpurun_out/pmc_c4; SE_AMD_NTT_PAIR=1 PMC_WL=c4 bash tools/gpu_run.sh pmc > gpurun_out/r5/pmc_c4_nttpair.log 2>&1; ls gpurun_out/pmc_c4
cp /tmp/new.so $L/libseal_embedded_amd.so
