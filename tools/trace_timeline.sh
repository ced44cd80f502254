#!/bin/bash
# gpurun -- 'bash tools/trace_timeline.sh <tag> [bench args]'
set -u
tag=${1:-t}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
W=gpurun_out/${tag}_w; rm -rf ${W}_trace
rocprofv3 --kernel-trace -d ${W}_trace -o run -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > /dev/null 2>&1
python3 tools/rocpd_timeline.py $(find ${W}_trace -name '*.db' | head -1)
rm -rf ${W}_trace
