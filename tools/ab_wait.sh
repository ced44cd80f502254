#!/bin/bash
run() { python bench.py --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.1f k' % (d['value']/1e3), end='')"; }
for rep in 1 2; do
echo "launch waits for the caller's stream:  20 steps $(UU3D_BENCH_WAIT_CALLER=1 run --steps 20 --warmup 5) | $(UU3D_BENCH_WAIT_CALLER=1 run --steps 20 --warmup 5) | $(UU3D_BENCH_WAIT_CALLER=1 run --steps 20 --warmup 5)   200 steps $(UU3D_BENCH_WAIT_CALLER=1 run --steps 200 --warmup 24)"
echo "launch without that wait:              20 steps $(run --steps 20 --warmup 5) | $(run --steps 20 --warmup 5) | $(run --steps 20 --warmup 5)   200 steps $(run --steps 200 --warmup 24)"
echo "  four slots, without the wait:        20 steps $(run --steps 20 --warmup 5 --streams 4) | $(run --steps 20 --warmup 5 --streams 4)   200 steps $(run --steps 200 --warmup 24 --streams 4)"
done
