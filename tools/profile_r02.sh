#!/bin/bash
# Runs on the GPU box (gpurun): kernel trace + separate --pmc passes (kernel-trace only, as the pool requires) of one bench
# command, reduced ON the box to small CSVs in gpurun_out/sum/ (the rocpd databases exceed what gpurun pulls back).
#   gpurun -- 'bash tools/profile_r02.sh <tag> [bench args...]'      e.g.  r02_a   |   r02_a_dense --config dense_351 --batch 32
set -u
tag=${1:-r02}; shift
args="$*"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/sum; rm -rf gpurun_out/${tag}_w_*
W=gpurun_out/${tag}_w
rocprofv3 --kernel-trace --stats -d ${W}_trace -o run -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline $args > gpurun_out/sum/${tag}_trace.log 2>&1
python3 tools/rocpd_summary.py stats gpurun_out/sum/${tag}_kernel_stats.csv ${W}_trace/*/run_results.db 2>/dev/null || python3 tools/rocpd_summary.py stats gpurun_out/sum/${tag}_kernel_stats.csv $(find ${W}_trace -name '*.db' | head -1)
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c -d ${W}_pmc_$n -o run -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph $args > gpurun_out/sum/${tag}_pmc_$n.log 2>&1
done
python3 tools/rocpd_summary.py pmc gpurun_out/sum/${tag}_pmc_summary.csv $(find ${W}_pmc_* -name '*.db')
if [ "${SQ:-1}" = "1" ]; then
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d ${W}_sq_$i -o run -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph $args > gpurun_out/sum/${tag}_sq_$i.log 2>&1
done
python3 tools/rocpd_counters.py gpurun_out/sum/${tag}_sq_summary.csv $(find ${W}_sq_* -name '*.db')
fi
rm -rf gpurun_out/${tag}_w_*
ls -la gpurun_out/sum/ | head -30
