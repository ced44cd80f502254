#!/bin/bash
run() { python bench.py --no-cpu-baseline --no-secondary --steps ${STEPS:-150} --warmup 16 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.1f k seq/s  %.4f ms/step' % (d['value']/1e3, d['ms_per_step']))"; }
echo "TIMING BUILD (UU3D_TC_LOO=256: chain transitions without memory traffic, results wrong)"
echo "batch 128 x 4 no chain: $(UU3D_TCHAIN=0 run)"
echo "batch 128 x 4 chain:    $(UU3D_TCHAIN=1 run)"
echo "batch 128 x 8 chain:    $(UU3D_TCHAIN=1 run --streams 8)"
echo "batch 512 x 4 no chain: $(UU3D_TCHAIN=0 run --batch 512)"
echo "batch 512 x 4 chain:    $(UU3D_TCHAIN=1 run --batch 512)"
echo "batch 128 x 4 no chain: $(UU3D_TCHAIN=0 run)"
