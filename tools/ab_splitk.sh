#!/bin/bash
run() { python bench.py --no-cpu-baseline --no-secondary --steps 200 --warmup 24 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.4f' % d['ms_per_step'], end='')"; }
for i in 1 2; do
echo "ms per step: target 768 $(run) | 384 $(UU3D_THR_SPLITK_TARGET=384 run) | 192 $(UU3D_THR_SPLITK_TARGET=192 run) | 96 $(UU3D_THR_SPLITK_TARGET=96 run) | 48 $(UU3D_THR_SPLITK_TARGET=48 run)"
done
