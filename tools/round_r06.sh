#!/bin/bash
# Round 6: EVERYTHING under profiles/r06/ that is a measurement comes from this one script on one GPU lease, and the
# figures of profiles/r06/README.md are generated from its files (tools/round_readme.py) -- no hand-copied numbers:
#   gpurun --timeout 3000 -- 'HGMM_COMMIT=<hash> bash tools/round_r06.sh'
# parity suite, smoke, the default bench line (+ legs), --mode pairs (batched; contexts x batch grid; round 5's one-call-per-
# pair form beside it), the N > 1 flows rehearsed on the one GPU, kernel trace of a batch of pairs, and -- from the SAME
# bench command -- rocprofv3 kernel statistics, the HBM traffic counters, the SQ counters of the batched tree kernels.
# Every profiled E-step run carries the store pacer's state (rate, steps, probes) in its own JSON line.
set -u
O=gpurun_out/r06
mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
grep -E "passed|failed|rc " $O/pytest_gpu.log | tail -3
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo "smoke rc $?"
HGMM_BENCH_LEGS_FILE=$O/bench_legs_n1.json timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc $? lines $(wc -l < $O/bench_n1.json)"
timeout 300 python bench.py --mode pairs > $O/bench_pairs_n1.json 2> $O/bench_pairs_n1.err; echo "pairs n1 rc $?"
for cfg in "1 1" "4 1" "8 1" "1 32" "2 32" "4 16" "4 32" "8 16" "8 32" "12 16"; do
  set -- $cfg
  timeout 200 python bench.py --mode pairs --contexts-per-gpu $1 --batch $2 --steps 20 --warmup 2 --min-time 3 --no-cpu-baseline \
    > $O/bench_pairs_n1_c$1_b$2.json 2> /dev/null; echo "pairs C=$1 B=$2 rc $?"
done
python tools/pair_batch_probe.py 1 4 16 32 64 > $O/pair_batch_probe.log 2>&1; echo "pair probe rc $?"
python tools/pair_batch_probe.py --f32 1 4 16 32 64 > $O/pair_batch_probe_f32.log 2>&1; echo "pair probe f32 rc $?"
python tools/pair_batch_probe.py --f32 --device-solve 1 32 > $O/pair_batch_probe_f32_device_solve.log 2>&1; echo "pair probe device solve rc $?"
HGMM_BENCH_DEVICE=0 timeout 300 python bench.py --mode pairs --gpus 2 --contexts-per-gpu 2 --no-cpu-baseline > $O/bench_pairs_n2_rehearsal_one_gpu.json 2> $O/bench_pairs_n2.err; echo "pairs rehearsal N=2 rc $?"
for N in 2 8; do
  HGMM_BENCH_DEVICE=0 HGMM_BENCH_LEGS_FILE=$O/bench_legs_n${N}_rehearsal.json timeout 600 python bench.py --gpus $N --collective ipc --steps 20 --warmup 5 > $O/bench_n${N}_rehearsal_one_gpu_peer_exchange.json 2> $O/bench_n${N}_ipc.err; echo "rehearsal ipc N=$N rc $?"
done
HGMM_BENCH_DEVICE=0 HGMM_BENCH_LEGS_FILE=$O/bench_legs_n2_rehearsal_host.json timeout 600 python bench.py --gpus 2 --collective host --steps 20 --warmup 5 > $O/bench_n2_rehearsal_one_gpu_host.json 2> $O/bench_n2_host.err; echo "rehearsal host N=2 rc $?"
# one profiled bench command: kernel statistics ...
timeout 900 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python bench.py --skip published_charts,replica_pairs > $O/bench_n1_under_rocprofv3.json 2> $O/bench_rocprof.err; echo "rocprof bench rc $?"
# ... and its HBM traffic counters (separate passes, as the guide prescribes), each pass's own JSON line kept (store pacer state)
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $C -d $O/pmc_$C -o pmc --output-format csv -- python bench.py --steps 4 --warmup 1 --estep-reps 3 --no-cpu-baseline --skip bunny,hgmm,tree_1M,fullcov,kmeans_init,registration,collective,published_charts,replica_pairs > $O/pmc_$C.stdout 2> $O/pmc_$C.stderr; echo "pmc $C rc $?"
done
python tools/pmc_summary.py $O > /dev/null
KT=$(find $O/kt -name "*kernel_trace.csv" | head -1); [ -n "$KT" ] && python tools/estep_patterns.py $KT > $O/rocprofv3_estep_by_grid.txt 2>&1
# a batch of 32 scan pairs (float32 scans): kernel trace, the last build launch by launch, SQ counters of the forest
# kernels (means per kernel and VALU instructions per dispatch); the same trace for float64 scans
timeout 300 rocprofv3 --kernel-trace -d $O/kt_batch -o kt --output-format csv -- python tools/pair_batch_probe.py --f32 32 > $O/pair_batch_probe_under_rocprofv3.log 2>&1
python tools/trace_summary.py $O/kt_batch > $O/kernel_trace_batch32.txt 2>&1
python tools/forest_levels.py $O/kt_batch > $O/forest_levels_batch32_f32.txt 2>&1
rm -rf $O/kt_batch
timeout 300 rocprofv3 --kernel-trace -d $O/kt_batch -o kt --output-format csv -- python tools/pair_batch_probe.py 32 > /dev/null 2>&1
python tools/forest_levels.py $O/kt_batch > $O/forest_levels_batch32_f64.txt 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES -d $O/pmc_batch -o pmc --output-format csv -- python tools/pair_batch_probe.py --f32 32 > /dev/null 2>&1
python tools/pmc_kernel.py $O/pmc_batch 2>&1 | grep -A9 "forest_\|tree_reg\|tree_ll_estep\|tree_moments" > $O/pmc_sq_batch32.txt
for k in forest_ll_estep forest_estep forest_moments forest_reg_estep; do echo "== SQ_INSTS_VALU per dispatch, $k (last 130)"; python tools/pmc_dispatches.py $O/pmc_batch SQ_INSTS_VALU $k 130; done > $O/pmc_valu_per_dispatch_batch32_f32.txt 2>&1
# how busy the GPU is under the default pairs line (eight contexts x 32 pairs)
timeout 300 rocprofv3 --kernel-trace -d $O/kt_pairs -o kt --output-format csv -- python bench.py --mode pairs --steps 6 --warmup 2 --min-time 1.5 --no-cpu-baseline --no-other-dtype > $O/bench_pairs_under_rocprofv3.json 2> /dev/null
python tools/gpu_busy.py $O/kt_pairs 0.4 > $O/gpu_busy_pairs_default.txt 2>&1; python tools/trace_summary.py $O/kt_pairs | head -12 >> $O/gpu_busy_pairs_default.txt 2>&1
rm -rf $O/kt_pairs
# flat full-covariance EM: float64 tile vs float32 tile (kernel ms, accuracy, phase clocks)
timeout 300 python tools/fullcov_f32_probe.py 10 > $O/fullcov_f32_probe.log 2>&1; echo "fullcov probe rc $?"
python tools/round_readme.py $O > $O/README.md 2> $O/readme.err; echo "readme rc $?"
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +20M -delete
rm -rf $O/kt_batch $O/pmc_batch
du -sh $O
