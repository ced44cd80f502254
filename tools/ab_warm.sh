#!/bin/bash
run() { python bench.py --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.1f k' % (d['value']/1e3), end='')"; }
for w in 5 50 500 5 50 500; do
  echo "warmup $w, 20 steps: $(run --steps 20 --warmup $w) | $(run --steps 20 --warmup $w) | $(run --steps 20 --warmup $w)"
done
for k in 20 40 80 200 1000; do echo "warmup 5, $k steps: $(run --steps $k --warmup 5) | $(run --steps $k --warmup 5)"; done
