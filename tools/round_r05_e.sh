#!/bin/bash
# Round 5, lease E: the scan-pair workload -- phases of one pair, pairs/s with 1 - 4 contexts per GPU.
set -u
O=gpurun_out/r05
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/pair_probe.py > $O/pair_probe.log 2>&1; echo "probe rc $?"; cat $O/pair_probe.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
for C in 1 2 4 6 8; do
  timeout 300 python bench.py --mode pairs --contexts-per-gpu $C --no-cpu-baseline > $O/bench_pairs_n1_c$C.json 2> $O/bench_pairs_c$C.err; echo "pairs C=$C rc $? $(python -c "import json;d=json.load(open('$O/bench_pairs_n1_c$C.json'));print(d['value'], d['ms_per_step'], d['accuracy']['max_misalignment_after_mm'])")"
done
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/kt_pair -o kt -- python tools/pair_probe.py > $O/pair_probe_rocprof.log 2>&1
python tools/trace_summary.py $O/kt_pair --seq 60 > $O/kernel_trace_pair.txt 2>&1; head -30 $O/kernel_trace_pair.txt
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
