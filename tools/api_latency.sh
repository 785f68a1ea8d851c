#!/bin/bash
# single-ciphertext latency of the reference API entry (se_encrypt_seeded) for the C1/C2/C4 shapes
set -e
cd "$(dirname "$0")/.."
LIB=$PWD/seal-embedded_amd/lib
gcc -std=gnu11 -O2 examples/api_digest.c -Iinclude -L$LIB -lseal_embedded_amd -Wl,-rpath,$LIB -o /tmp/api_digest
D=/tmp/api_lat_keys; rm -rf $D; mkdir -p $D
python - <<PY
import sys; sys.path[:0]=["tests"]
import vectors as V
for n in (1024, 4096, 16384): V.secret_key(n).tofile("$D/sk_%d.dat" % n)
PY
for shape in "1024 1" "4096 3" "16384 6"; do SE_AMD_DATA_PATH=$D /tmp/api_digest $shape sym 50 | tail -1 | sed "s/^/n,np = $shape: /"; done
