#!/bin/bash
# Round 5, lease B: replica tests, the default bench line, --mode pairs (1 rank; 2 and 4 ranks rehearsed on the one GPU),
# then the round's own HBM-traffic counters (separate --pmc passes of the bench command) and kernel statistics.
#   gpurun --timeout 2400 -- 'HGMM_COMMIT=<hash> bash tools/round_r05_b.sh'
set -u
O=gpurun_out/r05
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_concurrent_contexts_gpu.py tests/test_flat_gpu.py -m gpu -q --timeout 600 -x > $O/pytest_b.log 2>&1; echo "tests rc $?"; tail -3 $O/pytest_b.log
HGMM_BENCH_LEGS_FILE=$O/bench_legs_n1.json timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc $? lines $(wc -l < $O/bench_n1.json) bytes $(wc -c < $O/bench_n1.json)"
timeout 300 python bench.py --mode pairs > $O/bench_pairs_n1.json 2> $O/bench_pairs_n1.err; echo "pairs n1 rc $?"; cat $O/bench_pairs_n1.json | head -c 1500; echo
for N in 2 4; do
  HGMM_BENCH_DEVICE=0 timeout 300 python bench.py --mode pairs --gpus $N --no-cpu-baseline > $O/bench_pairs_n${N}_rehearsal_one_gpu.json 2> $O/bench_pairs_n$N.err; echo "pairs rehearsal N=$N rc $?"
done
timeout 700 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python bench.py --skip published_charts > $O/bench_n1_under_rocprofv3.json 2> $O/bench_rocprof.err; echo "rocprof bench rc $?"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $C -d $O/pmc_$C -o pmc --output-format csv -- python bench.py --steps 4 --warmup 1 --estep-reps 3 --no-cpu-baseline --skip bunny,hgmm,tree_1M,fullcov,kmeans_init,registration,collective,published_charts > $O/pmc_$C.stdout 2> $O/pmc_$C.stderr; echo "pmc $C rc $?"
done
python tools/pmc_summary.py $O
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +20M -delete
du -sh $O
