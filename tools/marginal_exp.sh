#!/bin/bash
# gpurun -- 'bash tools/marginal_exp.sh'  : what each launch class costs the forward -- the bench with that class left out (UU3D_SKIP, results
# wrong, time only), four batches in flight and one at a time, alternating with the complete forward on one box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { UU3D_SKIP=$1 python3 bench.py --steps 100 --warmup 10 --streams $2 --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('skip=$1 ($3) streams=$2 ms_per_step', d['ms_per_step'])"; }
for st in 0 1; do
  run 0 $st "nothing"
  run 1 $st "spatial stack"
  run 128 $st "temporal chain launches (6)"
  run 16 $st "attention"
  run 144 $st "temporal chain + attention"
  run 2 $st "QKV + fc1 panel GEMMs of the strided blocks"
  run 4 $st "projections of the strided blocks"
  run 145 $st "spatial stack + temporal chain + attention"
  run 0 $st "nothing"
done
