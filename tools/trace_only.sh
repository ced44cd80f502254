#!/bin/bash
# Runs on the GPU box: one rocprofv3 kernel trace of a bench command, reduced to gpurun_out/sum/<tag>_kernel_stats.csv.
#   gpurun -- 'bash tools/trace_only.sh <tag> [bench args...]'
set -u
tag=${1:-t}; shift
args="$*"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/sum; W=gpurun_out/${tag}_w
rm -rf ${W}_trace
rocprofv3 --kernel-trace --stats -d ${W}_trace -o run -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline $args > gpurun_out/sum/${tag}_trace.log 2>&1
python3 tools/rocpd_summary.py stats gpurun_out/sum/${tag}_kernel_stats.csv $(find ${W}_trace -name '*.db' | head -1)
rm -rf ${W}_trace
head -${LINES_OUT:-40} gpurun_out/sum/${tag}_kernel_stats.csv
