#!/bin/bash
mkdir -p gpurun_out
run() { python bench.py --no-cpu-baseline --no-secondary --steps ${STEPS:-200} --warmup 20 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.1f k seq/s  %.4f ms/step' % (d['value']/1e3, d['ms_per_step']))"; }
for q in 6 8; do
  echo "chain    $q queues: $(GPU_MAX_HW_QUEUES=$q UU3D_PIPE_QUEUES=$q UU3D_TCHAIN=1 run)"
  echo "no chain $q queues: $(GPU_MAX_HW_QUEUES=$q UU3D_PIPE_QUEUES=$q UU3D_TCHAIN=1 run)"
done
echo "chain    4 queues: $(UU3D_TCHAIN=1 run)"
echo "no chain 4 queues: $(run)"
echo "chain    8 queues: $(GPU_MAX_HW_QUEUES=8 UU3D_PIPE_QUEUES=8 UU3D_TCHAIN=1 run)"
echo "chain    8 queues 16 slots: $(GPU_MAX_HW_QUEUES=8 UU3D_PIPE_QUEUES=8 UU3D_TCHAIN=1 run --streams 16)"
