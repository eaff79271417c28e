#!/bin/bash
# Regenerates the per-round profile artefacts on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/profile_round.sh r01'
# 1. rocprofv3 --kernel-trace --stats of the default bench command  -> kernel_stats.csv + bench line
# 2. rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes -> per-kernel mean KB
# Everything lands in gpurun_out/prof_<round>/ ; tools/pmc_summary.py condenses it.
set -u
ROUND=${1:-r01}
OUT=gpurun_out/prof_$ROUND
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt --output-format csv -- \
    python bench.py > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.stderr"
tail -1 "$OUT/bench_n1.json" > "$OUT/bench_line.json"
for C in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --pmc $C -d "$OUT/pmc_$C" -o pmc --output-format csv -- \
        python bench.py --steps 4 --warmup 1 --estep-reps 3 --no-cpu-baseline > "$OUT/pmc_$C.stdout" 2> "$OUT/pmc_$C.stderr"
done
python tools/pmc_summary.py "$OUT"
ls -R "$OUT" | head -40
