#!/bin/bash
mkdir -p gpurun_out
run() { python bench.py --no-cpu-baseline --no-secondary --steps ${STEPS:-200} --warmup 24 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.1f k seq/s  %.4f ms/step' % (d['value']/1e3, d['ms_per_step']))"; }
echo "default:            $(run)"; python bench.py --no-cpu-baseline --no-secondary --steps 50 --warmup 8 2>&1 >/dev/null | grep -i "bench\]" | head -3
echo "--streams 8:        $(run --streams 8)"
echo "--streams 4:        $(run --streams 4)"
echo "--streams 12 (pool):$(run --streams 12)"
QUEUE_MAP_FIRST=1 QUEUE_MAP_R5=1 python tools/queue_map_exp.py --pool 32 --steps 200 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
echo "default:            $(run)"
echo "default, 20 steps:  $(run --steps 20 --warmup 5)"
