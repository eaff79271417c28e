#!/bin/bash
set -u
O=gpurun_out/r03_call13
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
tail -8 $O/pytest_gpu.log | cut -c1-300
timeout 400 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_call13/bench_n1.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], json.dumps(d['materialised_iteration'])[:400])
print({k:d[k].get('build_ms',d[k].get('ms_per_iteration')) for k in ('hgmm','tree_1M','fullcov')}, d['bunny']['gpu_it_per_s'])
PY
