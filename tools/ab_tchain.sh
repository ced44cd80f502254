#!/bin/bash
# gpurun -- 'bash tools/ab_tchain.sh'   the temporal chain against the round-4 launches under the pipelined bench, alternating on one box
mkdir -p gpurun_out
run() { python bench.py --no-cpu-baseline --no-secondary --steps ${STEPS:-200} --warmup 20 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.1f k seq/s  %.4f ms/step' % (d['value']/1e3, d['ms_per_step']))"; }
for rep in 1 2 3; do
  echo "chain    4 slots: $(UU3D_TCHAIN=1 run)"
  echo "no chain 4 slots: $(run)"
done
for s in 3 5 6 8; do echo "chain    $s slots: $(UU3D_TCHAIN=1 run --streams $s)"; done
echo "chain    4 slots, 20 steps: $(STEPS=20 UU3D_TCHAIN=1 run --warmup 5)"
echo "no chain 4 slots, 20 steps: $(STEPS=20 run --warmup 5)"
echo "chain    1 slot: $(UU3D_TCHAIN=1 run --streams 1)"
