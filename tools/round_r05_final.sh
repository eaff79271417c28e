#!/bin/bash
# Round 5: everything the round's GPU evidence consists of, on the final commit, in one lease:
#   gpurun --timeout 2700 -- 'HGMM_COMMIT=<hash> bash tools/round_r05_final.sh'
# parity suite, smoke, the default bench line (+ legs), --mode pairs (1 rank; 2 ranks rehearsed on the one GPU), the N > 1
# joint-fit flow rehearsed on one GPU, rocprofv3 kernel statistics + HBM traffic counters of the bench command.
set -u
O=gpurun_out/r05
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
grep -E "passed|failed|rc " $O/pytest_gpu.log | tail -3
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo "smoke rc $?"
HGMM_BENCH_LEGS_FILE=$O/bench_legs_n1.json timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc $? lines $(wc -l < $O/bench_n1.json) bytes $(wc -c < $O/bench_n1.json)"
timeout 300 python bench.py --mode pairs > $O/bench_pairs_n1.json 2> $O/bench_pairs_n1.err; echo "pairs n1 rc $?"
HGMM_BENCH_DEVICE=0 timeout 300 python bench.py --mode pairs --gpus 2 --contexts-per-gpu 2 --no-cpu-baseline > $O/bench_pairs_n2_rehearsal_one_gpu.json 2> $O/bench_pairs_n2.err; echo "pairs rehearsal N=2 rc $?"
for N in 2 8; do
  HGMM_BENCH_DEVICE=0 HGMM_BENCH_LEGS_FILE=$O/bench_legs_n${N}_rehearsal.json timeout 400 python bench.py --gpus $N --collective ipc --steps 20 --warmup 5 > $O/bench_n${N}_rehearsal_one_gpu_peer_exchange.json 2> $O/bench_n${N}_ipc.err; echo "rehearsal ipc N=$N rc $?"
done
HGMM_BENCH_DEVICE=0 HGMM_BENCH_LEGS_FILE=$O/bench_legs_n2_rehearsal_host.json timeout 400 python bench.py --gpus 2 --collective host --steps 20 --warmup 5 > $O/bench_n2_rehearsal_one_gpu_host.json 2> $O/bench_n2_host.err; echo "rehearsal host N=2 rc $?"
timeout 700 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python bench.py --skip published_charts,replica_pairs > $O/bench_n1_under_rocprofv3.json 2> $O/bench_rocprof.err; echo "rocprof bench rc $?"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $C -d $O/pmc_$C -o pmc --output-format csv -- python bench.py --steps 4 --warmup 1 --estep-reps 3 --no-cpu-baseline --skip bunny,hgmm,tree_1M,fullcov,kmeans_init,registration,collective,published_charts,replica_pairs > $O/pmc_$C.stdout 2> $O/pmc_$C.stderr; echo "pmc $C rc $?"
done
python tools/pmc_summary.py $O > /dev/null
KT=$(find $O/kt -name "*kernel_trace.csv" | head -1); [ -n "$KT" ] && python tools/estep_patterns.py $KT > $O/rocprofv3_estep_by_grid.txt 2>&1
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +20M -delete
du -sh $O
