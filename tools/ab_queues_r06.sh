#!/bin/bash
# Round 6: the temporal chain no longer touches memory between its stages; does the pipeline now want
# more than four hardware queues, or a different number of slots?  Alternating runs on one box.
run() { python bench.py --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.1f k (%d slots)' % (d['value']/1e3, d['config']['batches_in_flight']), end='')"; }
echo "4 queues, 8 slots (default):  20 steps $(run --steps 20 --warmup 5) | 200 steps $(run --steps 200 --warmup 24)"
for s in 4 6 12; do
  echo "4 queues, $s slots:            20 steps $(run --steps 20 --warmup 5 --streams $s) | 200 steps $(run --steps 200 --warmup 24 --streams $s)"
done
for q in 6 8; do
  echo "GPU_MAX_HW_QUEUES=$q, $q slots:   20 steps $(GPU_MAX_HW_QUEUES=$q UU3D_PIPE_QUEUES=$q run --steps 20 --warmup 5 --streams $q) | 200 steps $(GPU_MAX_HW_QUEUES=$q UU3D_PIPE_QUEUES=$q run --steps 200 --warmup 24 --streams $q)"
  echo "GPU_MAX_HW_QUEUES=$q, 2x slots:   20 steps $(GPU_MAX_HW_QUEUES=$q UU3D_PIPE_QUEUES=$q run --steps 20 --warmup 5) | 200 steps $(GPU_MAX_HW_QUEUES=$q UU3D_PIPE_QUEUES=$q run --steps 200 --warmup 24)"
done
echo "4 queues, 8 slots (default):  20 steps $(run --steps 20 --warmup 5) | 200 steps $(run --steps 200 --warmup 24)"
