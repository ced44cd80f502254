#!/bin/bash
# gpurun -- 'bash tools/ab_train_env.sh VAR "v1 v2 ..." [rounds]'  : the training-step bench with VAR set to each value (0 = unset), alternating
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
V=$1; VALS=${2:-"0 1"}; R=${3:-2}
for r in $(seq 1 $R); do
  for v in $VALS; do
    if [ "$v" = 0 ]; then unset $V; else export $V=$v; fi
    python3 bench.py --mode train --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$V=$v', d['value'], d['ms_per_step'])"
  done
done
