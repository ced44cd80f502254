#!/bin/bash
# Runs on the GPU box (gpurun): SQ wave-state counters of the forward's kernels, one rocprofv3 --pmc pass per counter group
# (kernel-trace only, as the pool requires), rocpd databases into gpurun_out/<tag>_sq_<group>.  Reduce locally with
#   python tools/rocpd_counters.py profiles/<tag>_sq_summary.csv gpurun_out/<tag>_sq_*/run_results.db
set -u
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/${tag}_sq_*
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d gpurun_out/${tag}_sq_$i -o run -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph > gpurun_out/${tag}_sq_$i.log 2>&1
done
ls gpurun_out/${tag}_sq_*/*
