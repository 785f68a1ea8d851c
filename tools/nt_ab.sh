#!/bin/bash
# Write amplification of the staged sampler at C4: two builds of the library (NT_LIBS="new prev": lib/prev.so against the product;
# round 4: whole-line stores of k_bulk_pair, and before that non-temporal accesses in k_ntt_fuse
# (lib/nt.so = make VARIANT=nt ABL=-DSEAMD_NTT_FUSE_NT): FETCH_SIZE / WRITE_SIZE per kernel (tools/gpu_run.sh pmc),
# then the step time of both builds alternating.  One gpurun call.
cd "$(dirname "$0")/.."
L=seal-embedded_amd/lib
cp $L/libseal_embedded_amd.so /tmp/new.so
for which in ${NT_LIBS:-new prev}; do
  if [ $which = new ]; then cp /tmp/new.so $L/libseal_embedded_amd.so; else cp $L/$which.so $L/libseal_embedded_amd.so; fi
  PMC_WL=c4 bash tools/gpu_run.sh pmc > /dev/null 2>&1
  rm -rf gpurun_out/pmc_c4_$which; mv gpurun_out/pmc_c4 gpurun_out/pmc_c4_$which
  python - <<PY
import csv, collections
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(float)
    cal = 0.0
    for row in csv.DictReader(open("gpurun_out/pmc_c4_$which/%s_counter_collection.csv" % ctr)):
        if row["Counter_Name"] == ctr:
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k] += float(row["Counter_Value"])
            if "copyBuffer" in k:
                cal = max(cal, float(row["Counter_Value"]))
    scale = (1 << 30) / (cal * 1024)
    for k, v in sorted(acc.items()):
        if k.startswith("seamd::"):
            print("$which", ctr, "%-40s %8.2f GB per step" % (k, v / 3 * 1024 * scale / 1e9))
PY
done
cp /tmp/new.so $L/libseal_embedded_amd.so
SE_BENCH_NO_CLOCK=1 AB_LIBS="${NT_LIBS:-new prev}" AB_WL="c4" bash tools/ab_bench.sh
