#!/bin/bash
# gpurun -- 'bash tools/ab_attn.sh VAR'  : dense-351 (batch 32) and the 117 / 200-token parity, attention launch time with VAR=1 / unset
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
V=${1:-UU3D_ATTN_NO_PIPE}
for r in 1 2 3; do for v in 1 0; do
  if [ $v = 1 ]; then export $V=1; else unset $V; fi
  python3 bench.py --config dense_351 --batch 32 --steps 40 --warmup 10 --streams 1 --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.readline()); a=d['roofline']['attention']; print('$V=$v', d['value'], 'attention', a)"
done; done
