#!/bin/bash
# first_multi_gpu.sh -- first contact with a box that shows MORE THAN ONE physical GPU.
# The N > 1 path (RCCL point-to-point gather of bench.py / sharding.py, hipMemcpyPeerAsync gather of se_multi.cpp) has
# only ever run on one physical device.  This script runs the existing pieces in order, each under a hard timeout, prints
# a one-line verdict per step and leaves every log under gpurun_out/first_multi_gpu/, so that a failure is localised
# to peer access / RCCL P2P / gather verification within ~10 GPU-minutes, and a SCALE-shaped line exists at the end.
#
#   bash tools/first_multi_gpu.sh            (from the repo root, on the multi-GPU box)
#   FMG_GPUS=4 bash tools/first_multi_gpu.sh (use 4 of the visible devices; default: all, at most 8)
cd "$(dirname "$0")/.."
OUT=gpurun_out/first_multi_gpu
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
SUMMARY=$OUT/summary.txt
: > $SUMMARY
say() { echo "$*" | tee -a $SUMMARY; }
step() {  # name timeout_s command...
  local name=$1 tmo=$2; shift 2
  local t0=$(date +%s)
  ( timeout $tmo "$@" ) > $OUT/$name.log 2>&1
  local rc=$?
  say "$(printf '%-28s rc=%-3s %4ss  %s' "$name" "$rc" "$(( $(date +%s) - t0 ))" "$( [ $rc = 0 ] && echo ok || ( [ $rc = 124 ] && echo TIMEOUT || echo FAILED ) )")"
  return $rc
}
last_json() { python - "$1" <<'PY'
import json, sys
line = None
for l in open(sys.argv[1], errors="replace"):
    l = l.strip()
    if l.startswith("{") and '"metric"' in l:
        line = l
if not line:
    print("   (no bench line)"); sys.exit(0)
d = json.loads(line)
g = d.get("gather") or {}
print("   n_gpus=%s value=%.4g %s ms_per_step=%.3f per_rank_ms=%s gather=%s verified=%s GB/s=%s note=%s" % (
    d.get("n_gpus"), d.get("value"), d.get("unit"), d.get("ms_per_step"),
    [x and round(x, 2) for x in (d.get("ranks") or {}).get("ms_per_step", [])] or None,
    g.get("form"), g.get("gather_verified"), g.get("GB/s") and round(g["GB/s"], 1), d.get("note")))
PY
}

NDEV=$(python - <<'PY'
import torch
buses = set()
for d in range(torch.cuda.device_count()):
    p = torch.cuda.get_device_properties(d)
    buses.add((p.pci_domain_id, p.pci_bus_id, p.pci_device_id))
print(len(buses))
PY
)
N=${FMG_GPUS:-$NDEV}; [ "$N" -gt 8 ] && N=8
say "distinct PCI devices visible: $NDEV (using $N)"
if [ "$NDEV" -lt 2 ] && [ -z "$FMG_FORCE" ]; then
  say "VERDICT: one physical device -- nothing to learn here (compute partition: $(rocm-smi --showcomputepartition 2>/dev/null | grep -i partition | head -1)); FMG_FORCE=1 runs the steps anyway (a dry run of this script: every rank lands on the one device)"
  exit 0
fi
[ "$NDEV" -lt 2 ] && N=${FMG_GPUS:-2}    # dry run on one device: two ranks share it

# 1. topology: are the devices xGMI peers at all?
step topo 60 rocm-smi --showtopo
grep -E "XGMI|PCIE" $OUT/topo.log | head -12 | sed 's/^/   /' | tee -a $SUMMARY

# 2. the distinct-device pytest cases (peer copies of se_multi.cpp, RCCL gather of sharding.py, vs the oracle)
step pytest_multi 900 python -m pytest tests/test_gpu_multi.py -m gpu -q -x --timeout=600
tail -3 $OUT/pytest_multi.log | sed 's/^/   /' | tee -a $SUMMARY

# 3. two ranks, BASELINE C2, with the timed + verified gather: RCCL P2P between two devices
step bench_c2_2gpu 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 \
     bench.py --gpus 2 --workload c2 --steps 5 --warmup 2 --others none --no-cpu-baseline
last_json $OUT/bench_c2_2gpu.log | tee -a $SUMMARY

# 4. all devices, C4 (the sharded BASELINE config): resident measurement only, then with the gather
step bench_c4_nogather 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29542 \
     bench.py --gpus $N --workload c4 --steps 5 --warmup 2 --others none --no-cpu-baseline --no-gather
last_json $OUT/bench_c4_nogather.log | tee -a $SUMMARY
step bench_c4_gather 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29543 \
     bench.py --gpus $N --workload c4 --steps 5 --warmup 2 --others none --no-cpu-baseline
last_json $OUT/bench_c4_gather.log | tee -a $SUMMARY

# 5. the native C entry (one process, one context + worker per device, hipMemcpyPeerAsync gather, self-verified)
step build_example 120 gcc examples/multi_device_encrypt.c -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ \
     -Lseal-embedded_amd/lib -lseal_embedded_amd -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/seal-embedded_amd/lib -Wl,-rpath,/opt/rocm/lib \
     -o $OUT/multi_device_encrypt \
  && step multi_device_encrypt 600 $OUT/multi_device_encrypt 4096 3 65536 \
  && grep -E "digest|gather_verified|distinct_devices|ct/s" $OUT/multi_device_encrypt.log | sed 's/^/   /' | tee -a $SUMMARY

# 6. the scaling curve the driver would record: N = 1, 2, 4, 8 back to back at C2
for n in 1 2 4 8; do
  [ $n -le $N ] || continue
  if [ $n = 1 ]; then
    step scale_c2_${n} 600 python bench.py --gpus 1 --workload c2 --steps 10 --warmup 3 --others none --no-cpu-baseline
  else
    step scale_c2_${n} 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29550 + n)) \
         bench.py --gpus $n --workload c2 --steps 10 --warmup 3 --others none --no-cpu-baseline --no-gather
  fi
  last_json $OUT/scale_c2_${n}.log | tee -a $SUMMARY
done
say "logs: $OUT/*.log"
