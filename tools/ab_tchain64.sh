#!/bin/bash
# NOTE: the library switches UU3D_TCHAIN64 / UU3D_TCHAIN16 this script alternates existed up to commit ed9e71b; the two earlier chain kernels
# now live under tools/ (uu3d_tchain.h, uu3d_tchain64.h) with their stand-alone harnesses.  Check that commit out to rerun this A/B.
# gpurun -- 'bash tools/ab_tchain64.sh': the pipelined bench with the 64-row chain (default) against the round-5 kernel (UU3D_TCHAIN64=0), alternating on one box
run() { python bench.py --no-cpu-baseline --no-secondary --steps ${STEPS:-200} --warmup 20 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.1f k seq/s  %.4f ms/step' % (d['value']/1e3, d['ms_per_step']))"; }
for rep in 1 2 3; do
  echo "64-row chain : $(UU3D_TCHAIN64=1 run)"
  echo "128-row chain: $(UU3D_TCHAIN64=0 run)"
done
echo "64-row chain , 20 steps: $(STEPS=20 UU3D_TCHAIN64=1 run --warmup 5)"
echo "128-row chain, 20 steps: $(STEPS=20 UU3D_TCHAIN64=0 run --warmup 5)"
echo "64-row chain , batch 512 x 4: $(UU3D_TCHAIN64=1 run --batch 512 --streams 4 --steps 50)"
echo "128-row chain, batch 512 x 4: $(UU3D_TCHAIN64=0 run --batch 512 --streams 4 --steps 50)"
echo "64-row chain , h36m_81 b256 x 4: $(UU3D_TCHAIN64=1 run --config h36m_81 --batch 256 --streams 4)"
echo "128-row chain, h36m_81 b256 x 4: $(UU3D_TCHAIN64=0 run --config h36m_81 --batch 256 --streams 4)"
