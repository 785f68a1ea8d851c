#!/bin/bash
# The reference's own benchmark programs (device/bench/*.c, built by `make -C oracle refbench` against the
# product library) on the GPU box: single-call latencies of the lower surface as the reference's timers
# report them.  usage: bash tools/ref_bench.sh > gpurun_out/ref_bench.log
cd "$(dirname "$0")/.."
ROOT=$PWD
W=/tmp/ref_bench_work; rm -rf $W; mkdir -p $W/adapter_output_data
python - <<PY
import sys; sys.path[:0] = ["$ROOT", "$ROOT/tests"]
import hashlib
import vectors as V
import __graft_entry__ as ge
pkg = ge.load_package()
for n in (1024, 2048, 4096, 8192, 16384):
    V.secret_key(n).tofile("$W/adapter_output_data/sk_%d.dat" % n)
ctx = pkg.Context(4096, 3)
pk0, pk1 = ctx.gen_public_key(V.secret_key(4096), hashlib.shake_256(b"golden-pk").digest(64), hashlib.shake_256(b"golden-ep").digest(64))
for j, q in enumerate(ctx.moduli()):
    pk0[j].tofile("$W/adapter_output_data/pk0_ntt_4096_%d.dat" % q); pk1[j].tofile("$W/adapter_output_data/pk1_ntt_4096_%d.dat" % q)
PY
cd $W
for b in sym asym ifft ntt uniform ternary cbd; do
  echo "== $b"
  timeout 300 $ROOT/oracle/_ref/ref_bench_gpu $b 2>&1 | grep -iE "avg|runtime|average|us\b" | tail -8
done
