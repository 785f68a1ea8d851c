export TMPDIR=/tmp
mkdir -p gpurun_out/r5
L=seal-embedded_amd/lib
cp $L/libseal_embedded_amd.so /tmp/new.so
for v in new pairnostore; do
  [ $v = new ] || cp $L/$v.so $L/libseal_embedded_amd.so
  for f in $((32768+65536)) 32768; do
  cd /tmp && rm -rf /tmp/prof && ( SE_BENCH_SKIP_STATUS=1 SE_BENCH_DEBUG_FLAGS=$f timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --workload c2 --no-cpu-baseline --others none ) > /tmp/p.log 2>&1
  cd $GRAFT_REPO_ROOT; echo "== $v flags=$f"; grep -h "k_bulk_lane_sync\|k_candidates\|k_sample_cbd" /tmp/prof/*/*kernel_stats.csv /tmp/prof/*kernel_stats.csv 2>/dev/null | cut -d, -f1-8 | cut -c1-200
  done
done > gpurun_out/r5/pair_nostore.log 2>&1
cp /tmp/new.so $L/libseal_embedded_amd.so
cat gpurun_out/r5/pair_nostore.log
