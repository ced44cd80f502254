#!/bin/bash
# gpurun -- 'bash tools/ab_panel8.sh'  : the forward with the 4-wave / 8-wave row-panel GEMM, alternating, one process each
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for r in 1 2 3; do
  for v in 1 0; do
    for st in 0 1; do
      UU3D_PANEL4=$v python3 bench.py --steps 100 --warmup 10 --streams $st --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('PANEL4=$v streams=$st', d['value'], d['ms_per_step'])"
    done
  done
done | tee gpurun_out/ab_panel8.txt
