#!/bin/bash
# Runs on the GPU box (gpurun): kernel-trace/stats and separate --pmc passes of the bench command, rocpd databases into
# gpurun_out/<tag>_*.  Post-process locally with tools/rocpd_summary.py and copy the CSVs into profiles/.
#   gpurun -- 'bash tools/profile_round.sh r01_final'
set -u
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/${tag}_*
rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_halves_trace -o run -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --halves > gpurun_out/${tag}_halves_trace.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_unsplit_trace -o run -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-halves > gpurun_out/${tag}_unsplit_trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  rocprofv3 --pmc $c -d gpurun_out/${tag}_unsplit_pmc_$c -o run -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --no-halves > gpurun_out/${tag}_unsplit_pmc_$c.log 2>&1
done
ls gpurun_out/${tag}_*/*
# the databases of one round exceed the 64 MiB gpurun pulls back: reduce them on the box (tools/rocpd_summary.py, tools/rocpd_counters.py)
# into gpurun_out/sum/ and delete the gpurun_out/${tag}_* directories before the call ends
