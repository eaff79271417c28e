#!/bin/bash
set -u
O=gpurun_out/r03_call11
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fullcov_gpu.py tests/test_multirank_gpu.py -m gpu -q --timeout 600 > $O/pytest_fullcov.log 2>&1; echo "pytest rc $?" >> $O/pytest_fullcov.log
tail -6 $O/pytest_fullcov.log | cut -c1-250
HGMM_FT_DEBUG=1 timeout 120 python tools/fullcov_prof.py 3 > $O/fullcov16_phase_clocks.log 2>&1
tail -33 $O/fullcov16_phase_clocks.log | head -16
timeout 120 python tools/fullcov_prof.py 10 2>&1 | tail -1
HGMM_FULLCOV_WAVES=8 timeout 120 python tools/fullcov_prof.py 10 2>&1 | tail -1
HGMM_FULLCOV_WAVES=8 HGMM_FT_DEBUG=1 timeout 120 python tools/fullcov_prof.py 3 > gpurun_out/r03_call11/fullcov8_phase_clocks.log 2>&1
tail -17 gpurun_out/r03_call11/fullcov8_phase_clocks.log | head -8
