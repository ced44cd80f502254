#!/bin/bash
# gpurun -- 'bash tools/batch_sweep.sh "128 153 ..."'  : forward throughput against the batch size (four batches in flight; one at a time)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for r in 1 2; do
for b in ${1:-128 136 144 153 160 170 192 256}; do
  for st in 0 1; do
    python3 bench.py --batch $b --steps 100 --warmup 10 --streams $st --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('batch $b streams=$st', d['value'], d['ms_per_step'])"
  done
done
done
