#!/bin/bash
# gpurun -- 'bash tools/ab_env.sh VAR [rounds]'  : the bench (four batches in flight, and one at a time) with VAR=1 / unset, alternating
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
V=${1:-UU3D_NO_LN_TAIL}; R=${2:-3}
for r in $(seq 1 $R); do
  for v in 1 0; do
    for st in 0 1; do
      if [ $v = 1 ]; then export $V=1; else unset $V; fi
      python3 bench.py --steps 100 --warmup 10 --streams $st --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$V=$v streams=$st', d['value'], d['ms_per_step'])"
    done
  done
done
