#!/bin/bash
run() { python bench.py --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.1f' % (d['value']/1e3), end='')"; }
for i in 1 2 3 4; do
echo "default: 20 steps $(run --steps 20 --warmup 5) $(run --steps 20 --warmup 5) | 200 steps $(run --steps 200 --warmup 24)    UU3D_TAIL=1: 20 steps $(UU3D_TAIL=1 run --steps 20 --warmup 5) $(UU3D_TAIL=1 run --steps 20 --warmup 5) | 200 steps $(UU3D_TAIL=1 run --steps 200 --warmup 24)"
done
