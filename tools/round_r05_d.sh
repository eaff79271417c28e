#!/bin/bash
# Round 5, lease D: grid-barrier probe (k-means++ design question), store-pacer recovery + stray-component tests, replica pool tests.
set -u
O=gpurun_out/r05
mkdir -p $O
export TMPDIR=/tmp
timeout 120 tools/gridbarrier > $O/gridbarrier.log 2>&1; echo "gridbarrier rc $?"; cat $O/gridbarrier.log
timeout 900 python -m pytest tests/test_flat_gpu.py tests/test_concurrent_contexts_gpu.py -m gpu -q --timeout 600 -s -k "pacer or stray or far_from or replica or fit_frames" > $O/pytest_d.log 2>&1; echo "tests rc $?"; grep -E "store pacer|\.\.\. and|stray|passed|failed|Error" $O/pytest_d.log | tail -12
