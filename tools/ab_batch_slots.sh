#!/bin/bash
# Round 5 (chain default on): sequences/s over batch size x slots in flight -- picks run_eval's default depth per batch size.
mkdir -p gpurun_out
run() { python bench.py --no-cpu-baseline --no-secondary --steps ${STEPS:-120} --warmup 16 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.1f k seq/s  %.4f ms/step' % (d['value']/1e3, d['ms_per_step']))"; }
echo "batch 128 x 8: $(run --steps 200)"
for b in 64 256 512 1024; do for s in 2 4 8; do
  echo "batch $b x $s: $(run --batch $b --streams $s --steps $((15360 / b)))"
done; done
echo "batch 128 x 8: $(run --steps 200)"
