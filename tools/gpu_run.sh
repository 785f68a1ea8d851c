#!/bin/bash
# usage: bash tools/gpu_run.sh [smoke] [ubench] [ubench2] [tests] [bench] [prof]   (run on the GPU box via gpurun)
mkdir -p gpurun_out
export TMPDIR=/tmp
for what in "$@"; do
case $what in
smoke) ( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log; tail -3 gpurun_out/smoke.log;;
ubench) ( timeout 120 ./tools/ubench ) > gpurun_out/ubench.log 2>&1; cat gpurun_out/ubench.log;;
ubench2) ( timeout 120 ./tools/ubench2 ) > gpurun_out/ubench2.log 2>&1; cat gpurun_out/ubench2.log;;
tests) ( timeout 2400 python -m pytest tests -m gpu -q --tb=short --timeout=900 ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -60 gpurun_out/pytest_gpu.log;;
bench) ( timeout 900 python bench.py ) > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log; tail -3 gpurun_out/bench.log;;
dist1) ( timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --batch 16384 --no-cpu-baseline --others none ) > gpurun_out/dist1.log 2>&1; tail -1 gpurun_out/dist1.log | cut -c1-400; ( timeout 600 python bench.py --gpus 8 --steps 3 --warmup 1 --batch 8192 --no-cpu-baseline --others none ) > gpurun_out/dist1b.log 2>&1; tail -1 gpurun_out/dist1b.log | cut -c1-400; ( timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 3 --warmup 1 --batch 8192 --no-cpu-baseline --others none ) > gpurun_out/dist1c.log 2>&1; tail -1 gpurun_out/dist1c.log | cut -c1-600;;
benchc5) for w in c5; do ( timeout 900 python bench.py --steps 5 --warmup 1 --workload $w --no-cpu-baseline ) > gpurun_out/bench_$w.log 2>&1; tail -1 gpurun_out/bench_$w.log; done;;
benchc3) for w in c3; do ( timeout 900 python bench.py --steps 5 --warmup 1 --workload $w --no-cpu-baseline ) > gpurun_out/bench_$w.log 2>&1; tail -1 gpurun_out/bench_$w.log; done;;
benchall) for w in c3 c5 c4 c1; do ( timeout 900 python bench.py --steps 5 --warmup 1 --workload $w --no-cpu-baseline ) > gpurun_out/bench_$w.log 2>&1; tail -1 gpurun_out/bench_$w.log; done;;
prof) cd /tmp && rm -rf /tmp/prof && ( timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o ${PROF_TAG:-r02} -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline ${PROF_ARGS:-} ) > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/prof; find /tmp/prof -name "*.csv" -size -2000k -exec cp {} gpurun_out/prof/ \; ; ls gpurun_out/prof; tail -3 gpurun_out/prof.log; for f in gpurun_out/prof/*kernel_stats*; do head -12 $f; done;;
pmc) W=${PMC_WL:-c2}; for c in FETCH_SIZE WRITE_SIZE; do rm -rf /tmp/pmc_$c; ( cd /tmp && timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o pmc -- python $GRAFT_REPO_ROOT/tools/pmc_run.py $W ) > gpurun_out/pmc_${W}_$c.log 2>&1; mkdir -p gpurun_out/pmc_$W; find /tmp/pmc_$c -name "*counter_collection.csv" -exec cp {} gpurun_out/pmc_$W/${c}_counter_collection.csv \; ; tail -2 gpurun_out/pmc_${W}_$c.log; done; ls -la gpurun_out/pmc_$W;;
sq) W=${PMC_WL:-c2}; i=0; for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"; do i=$((i+1)); rm -rf /tmp/sq_$i; ( cd /tmp && timeout 900 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/sq_$i -o sq -- python $GRAFT_REPO_ROOT/tools/pmc_run.py $W ) > gpurun_out/sq_${W}_$i.log 2>&1; mkdir -p gpurun_out/sq_$W; find /tmp/sq_$i -name "*counter_collection.csv" -exec cp {} gpurun_out/sq_$W/set${i}_counter_collection.csv \; ; tail -1 gpurun_out/sq_${W}_$i.log; done; ls -la gpurun_out/sq_$W;;
hostrate) ( timeout 600 python tools/host_rate.py ) > gpurun_out/host_rate.log 2>&1; cat gpurun_out/host_rate.log;;
ablate3) ( timeout 900 python tools/ablate3.py ) > gpurun_out/ablate3.log 2>&1; cat gpurun_out/ablate3.log;;
ablate2) ( timeout 600 python tools/ablate2.py ) > gpurun_out/ablate2.log 2>&1; cat gpurun_out/ablate2.log;;
esac
done
