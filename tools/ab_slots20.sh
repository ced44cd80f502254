#!/bin/bash
# the driver's shape (--steps 20 --warmup 5) and the steady state, by slots in flight
run() { python bench.py --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.1f k' % (d['value']/1e3), end='')"; }
for s in 4 8 12 4 8 12; do
  echo "slots $s: 20 steps $(run --steps 20 --warmup 5 --streams $s) | $(run --steps 20 --warmup 5 --streams $s) | $(run --steps 20 --warmup 5 --streams $s)   200 steps $(run --steps 200 --warmup 24 --streams $s)"
done
echo "no chain, slots 4: 20 steps $(UU3D_TCHAIN=0 run --steps 20 --warmup 5 --streams 4) | 200 steps $(UU3D_TCHAIN=0 run --steps 200 --warmup 24 --streams 4)"
echo "no chain, slots 8: 20 steps $(UU3D_TCHAIN=0 run --steps 20 --warmup 5 --streams 8) | 200 steps $(UU3D_TCHAIN=0 run --steps 200 --warmup 24 --streams 8)"
