#!/bin/bash
# What a round's GPU lease is spent on besides tools/profile_round.sh (run through gpurun from the repo root):
#   gpurun --timeout 2400 -- 'bash tools/round_checks.sh r03'
# parity suite, smoke, the default bench line, per-phase clocks of the full-covariance kernel, kernel traces of the two
# HGMM builds bench.py times, the issue-rate / overlap probes.  Everything lands in gpurun_out/checks_<round>/.
set -u
ROUND=${1:-r03}
O=gpurun_out/checks_$ROUND
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log | cut -c1-200
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo "smoke rc $?"
timeout 400 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc $?"
HGMM_FT_DEBUG=1 timeout 120 python tools/fullcov_prof.py 3 > $O/fullcov_phase_clocks.log 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/kt_c4 -o kt -- python tools/c4prof.py c4 4 > $O/c4prof.log 2>&1
python tools/trace_summary.py $O/kt_c4 --seq 40 > $O/kernel_trace_c4.txt 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/kt_t1m -o kt -- python tools/c4prof.py tree1m 2 > $O/t1mprof.log 2>&1
python tools/trace_summary.py $O/kt_t1m --seq 70 > $O/kernel_trace_tree1M.txt 2>&1
grep -h "build ms" $O/c4prof.log $O/t1mprof.log
timeout 60 python tools/c4prof.py both 5 2>&1 | tail -2
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
du -sh $O
