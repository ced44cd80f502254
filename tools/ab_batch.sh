#!/bin/bash
mkdir -p gpurun_out
run() { python bench.py --no-cpu-baseline --no-secondary --steps ${STEPS:-150} --warmup 16 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.1f k seq/s  %.4f ms/step' % (d['value']/1e3, d['ms_per_step']))"; }
echo "batch 128 x 4 no chain: $(run)"
for b in 256 384 512; do for s in 2 3 4; do
  echo "batch $b x $s chain:    $(UU3D_TCHAIN=1 run --batch $b --streams $s)"
  echo "batch $b x $s no chain: $(run --batch $b --streams $s)"
done; done
echo "batch 128 x 4 no chain: $(run)"
