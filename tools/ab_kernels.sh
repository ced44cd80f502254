#!/bin/bash
# gpurun -- 'bash tools/ab_kernels.sh ENVVAR'  : per-launch-class times (HIP events, eager) of the forward with ENVVAR=1 / 0, alternating
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
V=${1:-UU3D_PANEL4}
for r in 1 2; do
  for v in 1 0; do
    env $V=$v python3 bench.py --steps 30 --warmup 5 --streams 1 --no-secondary --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_forward']
print('$V=$v', d['value'], 'sum_kernel_ms', d['sum_kernel_ms'], ' '.join(f'{n}={k[n]*1e3:.1f}' for n in list(k)[:14]))"
  done
done
