cd $GRAFT_REPO_ROOT
LIB=$PWD/seal-embedded_amd/lib
gcc -std=gnu11 -O2 examples/api_digest.c -Iinclude -L$LIB -lseal_embedded_amd -Wl,-rpath,$LIB -o /tmp/api_digest
D=/tmp/api_lat_keys; rm -rf $D; mkdir -p $D
python - <<PY
import sys; sys.path[:0]=["tests"]
import vectors as V
V.secret_key(4096).tofile("$D/sk_4096.dat")
PY
cd /tmp && rm -rf /tmp/lt && SE_AMD_DATA_PATH=$D rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/lt -o lt -- /tmp/api_digest 4096 3 sym 5 > /tmp/lt.log 2>&1
tail -2 /tmp/lt.log
python3 - <<PY
import csv,glob
ev=[]
for f in glob.glob("/tmp/lt/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0][-40:]))
for f in glob.glob("/tmp/lt/**/*memory_copy_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"COPY "+r.get("Direction","")+" "+r.get("Bytes","")))
ev.sort()
t0=ev[0][0]
for s,e,n in ev[-40:]:
    print(f"{(s-t0)/1e6:10.3f} {(e-s)/1e3:9.1f} us  {n}")
PY
