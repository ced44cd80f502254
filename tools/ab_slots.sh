#!/bin/bash
mkdir -p gpurun_out
run() { python bench.py --no-cpu-baseline --no-secondary --steps ${STEPS:-200} --warmup 24 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.1f k seq/s  %.4f ms/step' % (d['value']/1e3, d['ms_per_step']))"; }
echo "no chain 4 slots: $(run)"
echo "chain    4 slots: $(UU3D_TCHAIN=1 run)"
for s in 8 12 16; do
  echo "chain    $s slots: $(UU3D_TCHAIN=1 run --streams $s)"
  echo "no chain $s slots: $(run --streams $s)"
done
echo "no chain 4 slots: $(run)"
