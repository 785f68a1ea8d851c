export TMPDIR=/tmp
mkdir -p gpurun_out/r5
L=seal-embedded_amd/lib
cp $L/libseal_embedded_amd.so /tmp/new.so
for v in qalias28; do
  cp $L/$v.so $L/libseal_embedded_amd.so
  ( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --tb=line --timeout=600 -k "encrypt_sym_vs_oracle or encode_only_config5 or encode_vs_oracle or all_pipeline_shapes or declined_plaintexts or encrypt_asym_vs_oracle" ) > gpurun_out/r5/pytest_$v.log 2>&1; echo "$v pytest rc=$?"; tail -2 gpurun_out/r5/pytest_$v.log
done
cp /tmp/new.so $L/libseal_embedded_amd.so
AB_LIBS="new qalias28" AB_WL="c2 c5" bash tools/ab_bench.sh > gpurun_out/r5/ab_qalias28.log 2>&1; cat gpurun_out/r5/ab_qalias28.log
