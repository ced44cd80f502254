#!/bin/bash
run() { python bench.py --no-cpu-baseline --no-secondary --steps ${STEPS:-200} --warmup 20 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.1f k seq/s  %.4f ms/step' % (d['value']/1e3, d['ms_per_step']))"; }
for rep in 1 2 3; do
  echo "no chain 4 slots: $(UU3D_TCHAIN=0 run)"
  echo "chain    4 slots: $(UU3D_TCHAIN=1 run)"
  echo "chain    8 slots: $(UU3D_TCHAIN=1 run --streams 8)"
  echo "no chain 8 slots: $(UU3D_TCHAIN=0 run --streams 8)"
done
echo "chain    6 slots: $(UU3D_TCHAIN=1 run --streams 6)"
echo "chain   12 slots: $(UU3D_TCHAIN=1 run --streams 12)"
echo "chain    8 slots, 20 steps: $(STEPS=20 UU3D_TCHAIN=1 run --streams 8 --warmup 8)"
echo "chain    4 slots, 20 steps: $(STEPS=20 UU3D_TCHAIN=1 run --warmup 5)"
echo "no chain 4 slots, 20 steps: $(STEPS=20 UU3D_TCHAIN=0 run --warmup 5)"
