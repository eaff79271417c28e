#!/bin/bash
# Round 5, lease C: float32-pdf stop rule of the HGMM build -- parity tests, then the 1M-point build both ways.
set -u
O=gpurun_out/r05
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_tree_gpu.py -m gpu -q --timeout 600 -x -s -k "float32 or mirror or 1M" > $O/pytest_tree_f32.log 2>&1; echo "tests rc $?"; grep -E "float32 pdfs|n = |passed|failed|Error|error" $O/pytest_tree_f32.log | tail -12
timeout 300 python tools/tree_f32_probe.py > $O/tree_f32_probe.log 2>&1; echo "probe rc $?"; grep -E "build_ms|loglik_kernel_ms_total|executed_pairs|bitwise|max_relative|level_iterations" $O/tree_f32_probe.log
