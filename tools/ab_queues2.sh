#!/bin/bash
# more hardware queues, now that nothing in the step loop touches the caller's stream (round 5)
run() { python bench.py --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.1f k (%d slots)' % (d['value']/1e3, d['config']['batches_in_flight']), end='')"; }
echo "4 queues (default):           20 steps $(run --steps 20 --warmup 5) | 200 steps $(run --steps 200 --warmup 24)"
for q in 6 8; do
  echo "GPU_MAX_HW_QUEUES=$q, $q probed:  20 steps $(GPU_MAX_HW_QUEUES=$q UU3D_PIPE_QUEUES=$q run --steps 20 --warmup 5) | 200 steps $(GPU_MAX_HW_QUEUES=$q UU3D_PIPE_QUEUES=$q run --steps 200 --warmup 24)"
  echo "GPU_MAX_HW_QUEUES=$q, $q slots:   20 steps $(GPU_MAX_HW_QUEUES=$q UU3D_PIPE_QUEUES=$q run --steps 20 --warmup 5 --streams $q) | 200 steps $(GPU_MAX_HW_QUEUES=$q UU3D_PIPE_QUEUES=$q run --steps 200 --warmup 24 --streams $q)"
done
echo "UU3D_TAIL=1:                  20 steps $(UU3D_TAIL=1 run --steps 20 --warmup 5) | 200 steps $(UU3D_TAIL=1 run --steps 200 --warmup 24)"
echo "4 queues (default):           20 steps $(run --steps 20 --warmup 5) | 200 steps $(run --steps 200 --warmup 24)"
