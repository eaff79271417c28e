#!/bin/bash
# Round 5, lease A: the new concurrency / multi-rank tests, then the whole parity suite, then the predict-mode probe.
set -u
O=gpurun_out/r05
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_concurrent_contexts_gpu.py tests/test_multirank_gpu.py -m gpu -q --timeout 600 -x > $O/pytest_new.log 2>&1; echo "new tests rc $?"; tail -5 $O/pytest_new.log
timeout 300 python tools/predict_modes.py > $O/predict_modes.log 2>&1; echo "predict rc $?"; cat $O/predict_modes.log
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu_a.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu_a.log
grep -E "passed|failed|rc " $O/pytest_gpu_a.log | tail -3
