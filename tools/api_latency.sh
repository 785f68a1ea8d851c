#!/bin/bash
# single-ciphertext latency of the reference API entry (se_encrypt_seeded), symmetric and public-key
set -e
cd "$(dirname "$0")/.."
LIB=$PWD/seal-embedded_amd/lib
gcc -std=gnu11 -O2 examples/api_digest.c -Iinclude -L$LIB -lseal_embedded_amd -Wl,-rpath,$LIB -o /tmp/api_digest
D=/tmp/api_lat_keys; rm -rf $D; mkdir -p $D
python - <<PY
import sys; sys.path[:0]=[".", "tests"]
import vectors as V
import __graft_entry__ as ge
pkg = ge.load_package()
for n, npr in ((1024, 1), (4096, 3), (16384, 6)):
    sk = V.secret_key(n); sk.tofile("$D/sk_%d.dat" % n)
    ctx = pkg.Context(n, npr)
    pk0, pk1 = ctx.gen_public_key(sk, bytes(64), bytes(range(64)))
    for j, q in enumerate(ctx.moduli()):
        pk0[j].tofile("$D/pk0_ntt_%d_%d.dat" % (n, q)); pk1[j].tofile("$D/pk1_ntt_%d_%d.dat" % (n, q))
    ctx.close()
PY
for mode in sym asym; do for shape in "1024 1" "4096 3" "16384 6"; do SE_AMD_DATA_PATH=$D /tmp/api_digest $shape $mode 50 | tail -1 | sed "s/^/$mode n,np = $shape: /"; done; done
