#!/bin/bash
set -u
O=gpurun_out/r03_call9
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
tail -12 $O/pytest_gpu.log | cut -c1-300
timeout 400 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc $?"
HGMM_FT_DEBUG=1 timeout 120 python tools/fullcov_prof.py 3 > $O/fullcov_phase_clocks.log 2>&1
tail -17 $O/fullcov_phase_clocks.log | head -9
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/kt_c4 -o kt -- python tools/c4prof.py c4 4 > $O/c4prof.log 2>&1
python tools/trace_summary.py $O/kt_c4 --seq 40 > $O/kt_c4_summary.txt 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/kt_t1m -o kt -- python tools/c4prof.py tree1m 2 > $O/t1mprof.log 2>&1
python tools/trace_summary.py $O/kt_t1m --seq 70 > $O/kt_t1m_summary.txt 2>&1
grep -h "build ms" $O/c4prof.log $O/t1mprof.log
find $O -name "*.db" -delete
du -sh $O
