#!/bin/bash
# round 3, GPU call 1: parity at every bench size + diagnostics for the kernels VERDICT r2 names
set -u
O=gpurun_out/r03_call1
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
timeout 300 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc $?"
HGMM_FT_DEBUG=1 timeout 120 python tools/fullcov_prof.py 3 > $O/fullcov_phase_clocks.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY \
    -d $O/pmc_fullcov_a -o pmc --output-format csv -- python tools/fullcov_prof.py 4 > $O/pmc_fullcov_a.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VALU SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_LDS_UNALIGNED_STALL \
    -d $O/pmc_fullcov_b -o pmc --output-format csv -- python tools/fullcov_prof.py 4 > $O/pmc_fullcov_b.log 2>&1
python tools/pmc_kernel.py $O/pmc_fullcov_a full_fused > $O/pmc_fullcov_summary.txt 2>&1
python tools/pmc_kernel.py $O/pmc_fullcov_b full_fused >> $O/pmc_fullcov_summary.txt 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/kt_c4 -o kt -- python tools/c4prof.py c4 4 > $O/c4prof.log 2>&1
python tools/trace_summary.py $O/kt_c4 --seq 60 > $O/kt_c4_summary.txt 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/kt_t1m -o kt -- python tools/c4prof.py tree1m 2 > $O/t1mprof.log 2>&1
python tools/trace_summary.py $O/kt_t1m --seq 70 > $O/kt_t1m_summary.txt 2>&1
HGMM_BENCH_HOSTCOMM=hgmm_reh2 HGMM_BENCH_DEVICE=0 timeout 300 python bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_n2_rehearsal.json 2> $O/bench_n2_rehearsal.err; echo "n2 rc $?"
rm -rf $O/pmc_fullcov_a/*/*.db $O/kt_c4/*/*.db 2>/dev/null
du -sh $O
