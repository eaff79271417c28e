#!/bin/bash
# Round 4: everything the round's GPU evidence consists of, in one lease (run through gpurun from the repo root):
#   gpurun --timeout 2400 -- 'bash tools/round_r04.sh'
# parity suite, smoke, the default bench line, the N > 1 flow rehearsed on one GPU (peer exchange and host backends),
# wall-vs-kernel probe of every public call, rocprofv3 kernel statistics + HBM traffic counters of the bench command,
# kernel traces of the two HGMM builds.  Everything lands in gpurun_out/r04/.
set -u
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
grep -E "passed|failed|rc " $O/pytest_gpu.log | tail -3
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo "smoke rc $?"
timeout 500 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc $? lines $(wc -l < $O/bench_n1.json)"
for N in 2 8; do
  HGMM_BENCH_DEVICE=0 timeout 400 python bench.py --gpus $N --collective ipc --steps 20 --warmup 5 > $O/bench_n${N}_rehearsal_one_gpu_peer_exchange.json 2> $O/bench_n${N}_ipc.err; echo "rehearsal ipc N=$N rc $?"
done
HGMM_BENCH_DEVICE=0 timeout 400 python bench.py --gpus 2 --collective host --steps 20 --warmup 5 > $O/bench_n2_rehearsal_one_gpu_host.json 2> $O/bench_n2_host.err; echo "rehearsal host N=2 rc $?"
timeout 300 python tools/api_probe.py 2>&1 | grep -v "^Failed to converge\|^GPU GMM TRAIN\|^Log Likelihood\|^ [0-9]\|^$\|^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" > $O/api_probe.log; echo "api_probe rc $?"
timeout 700 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python bench.py > $O/bench_n1_under_rocprofv3.json 2> $O/bench_rocprof.err; echo "rocprof bench rc $?"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $C -d $O/pmc_$C -o pmc --output-format csv -- python bench.py --steps 4 --warmup 1 --estep-reps 3 --no-cpu-baseline --skip bunny,hgmm,tree_1M,fullcov,kmeans_init,registration,collective > $O/pmc_$C.stdout 2> $O/pmc_$C.stderr; echo "pmc $C rc $?"
done
python tools/pmc_summary.py $O
KT=$(find $O/kt -name "*kernel_trace.csv" | head -1); [ -n "$KT" ] && python tools/estep_patterns.py $KT > $O/rocprofv3_estep_by_grid.txt 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/kt_c4 -o kt -- python tools/c4prof.py c4 4 > $O/c4prof.log 2>&1
python tools/trace_summary.py $O/kt_c4 --seq 40 > $O/kernel_trace_c4.txt 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/kt_t1m -o kt -- python tools/c4prof.py tree1m 2 > $O/t1mprof.log 2>&1
python tools/trace_summary.py $O/kt_t1m --seq 70 > $O/kernel_trace_tree1M.txt 2>&1
grep -h "build ms" $O/c4prof.log $O/t1mprof.log
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +20M -delete
du -sh $O
