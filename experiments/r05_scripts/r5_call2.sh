export TMPDIR=/tmp
mkdir -p gpurun_out/r5
L=seal-embedded_amd/lib
cp $L/libseal_embedded_amd.so /tmp/new.so
for v in sym5 prepairs; do
  cp $L/$v.so $L/libseal_embedded_amd.so
  ( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --tb=line --timeout=600 -k "encrypt_sym_vs_oracle or encode_only_config5 or encode_vs_oracle or all_pipeline_shapes" ) > gpurun_out/r5/pytest_$v.log 2>&1; echo "$v pytest rc=$?"; tail -2 gpurun_out/r5/pytest_$v.log
done
cp /tmp/new.so $L/libseal_embedded_amd.so
AB_LIBS="new nopairs prepairs sym5" AB_WL="c2 c5" bash tools/ab_bench.sh > gpurun_out/r5/ab_fused1.log 2>&1; cat gpurun_out/r5/ab_fused1.log
